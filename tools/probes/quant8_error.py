"""What 8-bit FMT weights would cost in accuracy: one CFG evaluation of the oracle with every Linear weight
quantised per output channel - fp16 4.2e-4, int8 1.6e-2, fp8 e4m3 5.2e-2 rel-L2 against the 4e-3 limit of the 16-bit path.  CPU, ~1 min."""
import sys, time, torch
sys.path.insert(0,'/root/repo')
from tests.util import load_pkg
pkg=load_pkg()
from oracle import float_oracle as O
cfg=pkg.config.FmtConfig()
sd=pkg.weights.synth_fmt_state(cfg,seed=1)
torch.manual_seed(0)
L=cfg.num_frames_for_clip
g=torch.Generator().manual_seed(0)
x=torch.randn(1,L,cfg.dim_w,generator=g)
c=pkg.pipeline.synth_conditions(cfg,L,seed=0)
px=torch.zeros(1,cfg.num_prev_frames,cfg.dim_w)
def ev(s): return O.fmt_forward_cfv(s,cfg,torch.tensor([0.5]),x,c['wa'],c['r_s'],c['we'],px,px,None,2.0,1.0,1.0)
ref=ev(sd)
def q_fp8(w):  # e4m3 per output channel scale
    s=w.abs().amax(dim=1,keepdim=True).clamp_min(1e-12)/448.0
    return (w/s).to(torch.float8_e4m3fn).to(torch.float32)*s
def q_int8(w):
    s=w.abs().amax(dim=1,keepdim=True).clamp_min(1e-12)/127.0
    return torch.round(w/s).clamp(-127,127)*s
def q_fp16(w): return w.half().float()
for name,q in (('fp16',q_fp16),('fp8-e4m3/channel',q_fp8),('int8/channel',q_int8)):
    s2={k:(q(v) if (v.dim()==2 and k.endswith('weight')) else v) for k,v in sd.items()}
    out=ev(s2)
    print(name,'weights only: rel-L2 of one CFG evaluation %.3e'%float((out-ref).norm()/ref.norm()))
