"""Model-shape constants of the hot path (values are the checkpoint contract,
reference src/nodes/options/base_options.py:10-60)."""
from dataclasses import dataclass


@dataclass
class FmtConfig:
    dim_w: int = 512          # motion latent (base_options.py:37)
    dim_a: int = 512          # audio feature (base_options.py:36)
    dim_e: int = 7            # emotion classes (base_options.py:40)
    dim_h: int = 1024         # hidden (base_options.py:38)
    fmt_depth: int = 8        # base_options.py:41
    num_heads: int = 8        # base_options.py:42
    mlp_ratio: float = 4.0    # base_options.py:43
    num_prev_frames: int = 10     # base_options.py:45
    num_frames_for_clip: int = 50  # int(wav2vec_sec * fps), FMT.py:209
    attention_window: int = 2     # base_options.py:29

    @property
    def n_tokens(self):
        return self.num_prev_frames + self.num_frames_for_clip

    @property
    def head_dim(self):
        return self.dim_h // self.num_heads

    @classmethod
    def from_options(cls, opt):
        return cls(dim_w=opt.dim_w, dim_a=opt.dim_a, dim_e=opt.dim_e, dim_h=opt.dim_h,
                   fmt_depth=opt.fmt_depth, num_heads=opt.num_heads, mlp_ratio=opt.mlp_ratio,
                   num_prev_frames=int(opt.num_prev_frames),
                   num_frames_for_clip=int(opt.wav2vec_sec * opt.fps),
                   attention_window=int(opt.attention_window))


# Reduced shape used by CPU-side golden tests (same head_dim as the full model).
def small_fmt_config():
    return FmtConfig(dim_w=128, dim_a=128, dim_e=7, dim_h=256, fmt_depth=2, num_heads=2,
                     mlp_ratio=4.0, num_prev_frames=10, num_frames_for_clip=50, attention_window=2)
