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


@dataclass
class AudioConfig:
    """Shape of the audio conditioning encoder = the bundled wav2vec2_base config
    (reference src/nodes/model_configs/wav2vec2_base/config.json) + opt.dim_w / opt.only_last_features."""
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    dim_w: int = 512
    only_last_features: bool = False
    # wav2vec2-large family switches (the speech-emotion model, model_configs/emotion_ser/config.json)
    feat_extract_norm: str = "group"
    do_stable_layer_norm: bool = False
    conv_bias: bool = False
    num_labels: int = 0       # > 0: classification head instead of the audio projection

    def to_hf(self):
        """The same shape as a transformers Wav2Vec2Config (eager attention, group-norm feature extractor)."""
        from transformers import Wav2Vec2Config
        return Wav2Vec2Config(conv_dim=tuple(self.conv_dim), conv_kernel=tuple(self.conv_kernel), conv_stride=tuple(self.conv_stride),
                              num_feat_extract_layers=len(self.conv_dim), hidden_size=self.hidden_size,
                              num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                              intermediate_size=self.intermediate_size, num_conv_pos_embeddings=self.num_conv_pos_embeddings,
                              num_conv_pos_embedding_groups=self.num_conv_pos_embedding_groups, layer_norm_eps=self.layer_norm_eps,
                              feat_extract_norm=self.feat_extract_norm, conv_bias=self.conv_bias,
                              do_stable_layer_norm=self.do_stable_layer_norm, attn_implementation="eager",
                              **({"num_labels": self.num_labels} if self.num_labels else {}))

    @classmethod
    def from_hf(cls, hf, dim_w=512, only_last_features=False, num_labels=0):
        return cls(tuple(hf.conv_dim), tuple(hf.conv_kernel), tuple(hf.conv_stride), hf.hidden_size, hf.num_hidden_layers,
                   hf.num_attention_heads, hf.intermediate_size, hf.num_conv_pos_embeddings, hf.num_conv_pos_embedding_groups,
                   hf.layer_norm_eps, dim_w, only_last_features, getattr(hf, "feat_extract_norm", "group"),
                   bool(getattr(hf, "do_stable_layer_norm", False)), bool(getattr(hf, "conv_bias", False)), num_labels)


def small_audio_config():
    """Reduced wav2vec2 shape for fast parity tests (same head dim 64, LayerNorm widths multiples of 256)."""
    return AudioConfig(conv_dim=(256,) * 7, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=16, dim_w=256)


def emotion_audio_config():
    """The speech-emotion recogniser: wav2vec2-large-xlsr shape of model_configs/emotion_ser/config.json with 7 labels."""
    return AudioConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                       feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True, num_labels=7)


def small_emotion_config():
    return AudioConfig(conv_dim=(256,) * 7, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=16, feat_extract_norm="layer",
                       do_stable_layer_norm=True, conv_bias=True, num_labels=7)
