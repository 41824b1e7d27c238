cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_g; mkdir -p $O
python bench.py --seconds 60 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c2.json 2>/dev/null; python -c "import json;d=json.load(open('$O/c2.json'));print('60s', d['value'], d['ms_per_step'], d['stage_ms'], d['hot_path'])"
python bench.py --seconds 30 --dynamic-we --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/c4.json 2>/dev/null; python -c "import json;d=json.load(open('$O/c4.json'));print('30s dyn', d['value'], d['ms_per_step'], d['stage_ms'], d['hot_path'])"
python bench.py --seconds 1 --nfe 10 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/c0.json 2>/dev/null; python -c "import json;d=json.load(open('$O/c0.json'));print('1s nfe10', d['value'], d['ms_per_step'], d['stage_ms'], d['hot_path'])"
python bench.py --fmt-dtype bf16 --no-cpu-baseline --no-roofline > $O/bf16.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bf16.json'));print('bf16', d['value'], d['ms_per_step'], d['stage_ms'])"
python -m pytest tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_dec_gpu.py tests/test_enc_gpu.py -m gpu -q -s 2>&1 | grep -i "psnr\|rel-L2\|passed\|failed" | head -40
