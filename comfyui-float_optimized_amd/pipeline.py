"""The hot path as one object: (conditioning tensors in HBM) -> r_d -> frames.
Mirrors FLOAT.sample + decode_latent_into_processed_images (reference FLOAT.py:172-253, 113-169)
with the noise stream explicit."""
import math

import torch

from .config import FmtConfig
from .decoder import SynthesisHIP
from .fmt import FlowMatchingTransformerHIP, draw_noise


class FloatHotPath:
    def __init__(self, fmt_state, dec_state, cfg: FmtConfig = None, device="cuda:0", size=512, fmt_dtype="bf16",
                 dec_dtype="bf16", max_frames=16, use_graph=True):
        self.cfg = cfg or FmtConfig()
        self.device = torch.device(device)
        self.size = size
        self.fmt = FlowMatchingTransformerHIP(fmt_state, self.cfg, device, fmt_dtype, use_graph)
        self.dec = SynthesisHIP(dec_state, size, self.cfg.dim_w, device, dec_dtype, max_frames)

    def n_chunks(self, T):
        return int(math.ceil(T / self.cfg.num_frames_for_clip))

    @torch.no_grad()
    def sample(self, r_s, wa, we, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, seed=15, noise=None,
               include_r_cfg=False):
        """r_d (B,T,512).  `noise` (n_chunks,B,50,512) overrides the seeded CPU-generator draw."""
        if noise is None:
            noise = draw_noise(self.n_chunks(wa.shape[1]), wa.shape[0], self.cfg, seed)
        return self.fmt.sample(r_s, wa, we, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg)

    @torch.no_grad()
    def decode(self, s_r, feats, r_d, frame_range=None):
        """(T,H,W,3) fp32 in [0,1] on the GPU for batch item 0; frame_range=(t0,t1) decodes a shard."""
        if feats is not None:
            self.dec.set_feats(feats)
        rd = r_d[0] if r_d.dim() == 3 else r_d
        if frame_range is not None:
            rd = rd[frame_range[0]:frame_range[1]]
        return self.dec.decode_latent_into_processed_images(s_r, rd)

    @torch.no_grad()
    def generate(self, r_s, wa, we, s_r, feats, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, seed=15,
                 noise=None):
        r_d = self.sample(r_s, wa, we, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, seed, noise)
        return self.decode(s_r, feats, r_d)


def synth_conditions(cfg: FmtConfig, T, seed=0, dynamic_we=False, device="cpu"):
    """Synthetic stand-ins for the off-path encoders' outputs (formats of SURVEY.md section 8a):
    wa ~ SiLU(LayerNorm-ed projection) (FLOAT.py:338-342), we softmax scores, r_s, s_r."""
    g = torch.Generator().manual_seed(1000 + seed)
    wa = torch.nn.functional.silu(torch.randn(1, T, cfg.dim_a, generator=g))
    if dynamic_we:
        nwin = int(math.ceil(T / cfg.num_frames_for_clip))
        w = torch.softmax(torch.randn(1, nwin, cfg.dim_e, generator=g), -1)
        idx = torch.clamp((torch.arange(T).float() * nwin / T).long(), max=nwin - 1)  # nearest upsample, nodes_vadv.py:835-838
        we = w[:, idx]
    else:
        we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1)
    r_s = torch.randn(1, cfg.dim_w, generator=g) * 0.5
    s_r = torch.randn(1, cfg.dim_w, generator=g)
    return dict(wa=wa.to(device), we=we.to(device), r_s=r_s.to(device), s_r=s_r.to(device))
