import sys, torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
C, W = pkg.config, pkg.weights
cfg = C.small_fmt_config()
sd = W.synth_fmt_state(cfg, 9)
fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
T = 70
g = torch.Generator().manual_seed(0)
r_s = torch.randn(1, cfg.dim_w, generator=g).cuda()
wa = torch.randn(1, T, cfg.dim_a, generator=g).cuda()
we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1).cuda()
noise = pkg.fmt.draw_noise(2, 1, cfg, seed=15).cuda()
plain = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0).clone()
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0)
    torch.cuda.current_stream().synchronize()
    with torch.cuda.graph(graph, stream=side):
        captured = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
def rep():
    graph.replay(); torch.cuda.synchronize(); return captured.clone()
a = rep(); b = rep()
print("replay == plain:", torch.equal(a, plain), " replay deterministic:", torch.equal(a, b))
wa.mul_(0.5); torch.cuda.synchronize()
c1 = rep(); c2 = rep()
want = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0).clone(); torch.cuda.synchronize()
want2 = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0).clone(); torch.cuda.synchronize()
d = (c1 - want).abs().amax(dim=(0, 2))
print("after mul: replay deterministic:", torch.equal(c1, c2), " plain deterministic:", torch.equal(want, want2), " replay == plain:", torch.equal(c1, want),
      " differing frames:", int((d > 0).sum()), "first", torch.nonzero(d).flatten()[:3].tolist(), "max", float(d.max()))
c3 = rep()
print("replay after plain == earlier replay:", torch.equal(c3, c1), " == plain:", torch.equal(c3, want))
