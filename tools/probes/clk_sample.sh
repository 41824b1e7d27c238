# Shader clock and socket power DELIVERED while a workload runs (rocm-smi from outside, 5 samples 0.7 s apart after a 4-s lead):
#   sh tools/probes/clk_sample.sh "label" ENV=... command args...      (run from the repo root on the GPU box)
# The MFMA peak a kernel can be held to is 2.5 PFLOP/s x (delivered clock / 2.4 GHz); MI355X holds 2.4 GHz only on light loads.
label=$1; shift
echo "== $label"
env "$@" > /tmp/clk_sample_out.txt 2>&1 &
pid=$!
sleep ${LEAD:-4}
for i in 1 2 3 4 5; do
  c=$(rocm-smi --showclocks 2>&1 | grep -E "sclk" | sed 's/.*(\(.*\)).*/\1/')
  p=$(rocm-smi --showpower 2>&1 | grep -i -E "Power \(W\)" | sed 's/.*: //')
  echo "  sclk $c  power $p W"
  sleep 0.7
done
wait $pid
tail -${TAIL:-1} /tmp/clk_sample_out.txt
