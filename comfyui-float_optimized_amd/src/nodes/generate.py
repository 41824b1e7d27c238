"""InferenceAgent of the MI355X build (reference generate.py:84-173): owns the weights, the host-side
conditioning encoders and the HIP hot path, and exposes run_inference with the reference signature."""
import contextlib
import math
import os

import torch

from ... import host_models, weights
from ...audio import Audio2EmotionHIP, AudioEncoderHIP
from ...config import AudioConfig, FmtConfig, emotion_audio_config, small_audio_config, small_emotion_config
from ...encoder import EncoderHIP
from ...pipeline import FloatHotPath, report_range
from . import SYNTHETIC_MODEL, main_logger

# key prefixes of the unified checkpoint (utils/downloader.py:35-42)
PREFIXES = {
    "enc": "motion_autoencoder.enc.",
    "dec": "motion_autoencoder.dec.",
    "proj": "audio_encoder.audio_projection.",
    "fmt": "fmt.",
    "wav2vec": "audio_encoder.wav2vec2.",
    "ser": "emotion_encoder.wav2vec2_for_emotion.",
}


def split_unified(state):
    """FLOAT.state_dict() -> per-part dicts with the prefixes stripped."""
    parts = {k: {} for k in PREFIXES}
    for key, v in state.items():
        for part, pre in PREFIXES.items():
            if key.startswith(pre):
                parts[part][key[len(pre):]] = v
                break
    return parts


class InferenceAgent:
    def __init__(self, opt, parts=None, device=None, max_frames=32, use_graph=2, fmt_dtype=None, dec_dtype=None, aud_dtype=None):
        self.opt = opt
        self.rank = torch.device(device if device is not None else getattr(opt, "rank", "cuda:0"))
        self.cfg = FmtConfig.from_options(opt)
        if parts is None:
            parts = self._load_parts(opt)
        self.dec_sd = parts["dec"]
        # the host copy of the weights stays with the agent (the reference's offload device, nodes.py:139): offload() frees
        # every device allocation, to_target() rebuilds the operators from here
        self._parts = parts
        from ... import native as _native
        canon = _native.canon_dtype  # 'float16' and 'fp16' are one type to every later `== "fp16"` test
        self._build = dict(max_frames=max_frames, use_graph=use_graph,
                           fmt_dtype=canon(fmt_dtype or os.environ.get("FLOAT_AMD_FMT_DTYPE", "fp16")),
                           dec_dtype=canon(dec_dtype or os.environ.get("FLOAT_AMD_DEC_DTYPE", "fp16")),
                           aud_dtype=canon(aud_dtype or os.environ.get("FLOAT_AMD_AUD_DTYPE", "fp16")))
        self.G = None
        self.to_target()

    # ------------------------------------------------------------------ residency (reference: model_to_target, nodes.py:173-175)
    def to_target(self):
        """Build the HIP operators on self.rank from the host weights if they are not resident (no-op otherwise)."""
        if self.G is not None:
            return self
        opt, parts, b = self.opt, self._parts, self._build
        # 16-bit MFMA operand types (fp32 accumulation): fp16 in both operators gives ~8x lower rounding error than
        # bf16 at the same rate (end-to-end 48.7 vs 34.1 dB on BASELINE configs[0]); FLOAT_AMD_FMT_DTYPE=bf16
        # selects the type BASELINE configs[1] names.
        self.G = FloatHotPath(parts["fmt"], parts["dec"], self.cfg, self.rank, opt.input_size, fmt_dtype=b["fmt_dtype"],
                              dec_dtype=b["dec_dtype"], max_frames=b["max_frames"], use_graph=b["use_graph"])
        self.G.fmt.set_method(getattr(opt, "torchdiffeq_ode_method", "euler"))
        # appearance encoder + Encoder.fc + Direction as one HIP operator (float_enc_*); same 16-bit type as the
        # decoder so the skip features go to it without an fp32 round trip
        self.enc = EncoderHIP(parts["enc"], opt.input_size, opt.dim_w, getattr(opt, "dim_m", 20), self.rank,
                              dtype=self.G.dec.dtype, direction_weight=parts["dec"]["direction.weight"])
        # wav2vec2 + audio projection as one HIP operator (float_aud_*)
        aud_sd, aud_cfg = parts["audio_encoder"]
        self.audio_encoder = AudioEncoderHIP(aud_sd, aud_cfg, self.rank, dtype=b["aud_dtype"], sampling_rate=opt.sampling_rate, fps=opt.fps)
        # speech-to-emotion (emotion="none"): the wav2vec2-large variant of the same operator with its classification head
        ser = parts.get("emotion_encoder")
        self.emotion_encoder = Audio2EmotionHIP(ser[0], ser[1], self.rank, dtype=b["aud_dtype"]) if ser is not None else None
        # callable(a) -> (1,7) softmax scores; None disables emotion="none"
        self.emotion_predictor = self.emotion_encoder.predict_emotion if ser is not None else parts.get("emotion_predictor")
        return self

    def offload(self):
        """Free every device allocation of the agent (packed weights, workspaces, hipGraphs, staging and cached tensors:
        ~6.5 GB at the default shapes) - the counterpart of the reference moving G back to the offload device after a node
        call.  The next call rebuilds the operators from the host weights (to_target); results are bitwise the same."""
        if self.G is None:
            return
        torch.cuda.current_stream(self.rank).synchronize()
        for op in [self.G.fmt, self.G.dec, self.enc, self.audio_encoder, self.emotion_encoder] + list(self.G.__dict__.get("_fmt_batched", {}).values()):
            if op is not None:
                op.close()
        if self.emotion_encoder is not None:
            self.emotion_predictor = None
        self.G = self.enc = self.audio_encoder = self.emotion_encoder = None
        self.__dict__.pop("_we_cache", None)
        self.__dict__.pop("_noise_pin", None)
        self.__dict__.pop("_noise_pin_b", None)
        self.__dict__.pop("_feat_slots", None)
        with torch.cuda.device(self.rank):
            torch.cuda.empty_cache()

    @property
    def resident(self):
        return self.G is not None

    @contextlib.contextmanager
    def model_to_target(self, offload_after=None):
        """The reference wraps every node call in `with model_to_target(logger, float_pipe.G)` (nodes.py:173-175): operators
        on the target device inside, back on the offload device after.  Here staying resident is the default (288 GB of
        HBM; a rebuild costs seconds of host packing): offload_after=True, or FLOAT_AMD_OFFLOAD=always, gives the reference's
        behaviour; FLOAT_AMD_OFFLOAD=never (default) keeps the operators."""
        self.to_target()
        try:
            yield self
        finally:
            if offload_after if offload_after is not None else os.environ.get("FLOAT_AMD_OFFLOAD", "never").lower() == "always":
                self.offload()

    # ------------------------------------------------------------------ weights
    @staticmethod
    def _load_parts(opt):
        path = getattr(opt, "ckpt_path", None)
        if path is None or os.path.basename(path) == SYNTHETIC_MODEL or os.environ.get("FLOAT_AMD_SYNTHETIC") == "1":
            return InferenceAgent.synthetic_parts(opt)
        if not os.path.exists(path):
            raise FileNotFoundError("Checkpoint file not found: %s" % path)
        from safetensors.torch import load_file
        parts = split_unified(load_file(path, device="cpu"))
        aud_sd = {"wav2vec2." + k: v for k, v in parts["wav2vec"].items()}
        aud_sd.update({"audio_projection." + k: v for k, v in parts["proj"].items()})
        parts["audio_encoder"] = (aud_sd, AudioConfig(dim_w=opt.dim_w, only_last_features=opt.only_last_features))
        if parts["ser"]:
            parts["emotion_encoder"] = (parts["ser"], emotion_audio_config())
        return parts

    @staticmethod
    def synthetic_parts(opt, seed=0):
        """Seeded random weights in the checkpoint layout (no network / no checkpoint available)."""
        cfg = FmtConfig.from_options(opt)
        acfg = small_audio_config()
        acfg.dim_w = opt.dim_w
        return dict(enc=weights.synth_encoder_state(opt.input_size, seed=seed), dec=weights.synth_decoder_state(opt.input_size, seed=seed),
                    fmt=weights.synth_fmt_state(cfg, seed=seed),
                    audio_encoder=(weights.synth_audio_state(acfg, seed=seed), acfg),
                    emotion_encoder=(weights.synth_audio_state(small_emotion_config(), seed=seed), small_emotion_config()))

    # ------------------------------------------------------------------ inference
    def _one_hot(self, emo):
        """(1,1,7) one-hot of a label on the device, made once per label (a fresh one costs a blocking scalar upload per clip)."""
        cache = self.__dict__.setdefault("_we_cache", {})
        idx = host_models.emotion_index(emo)
        if idx not in cache:
            cache[idx] = host_models.emotion_one_hot(emo, "cpu").to(self.rank)
        return cache[idx]

    def _noise_batch_to_device(self, n_chunks, seeds):
        """(n_chunks, B, L, W): item i draws its own sequential stream from seeds[i] (what infer_device draws for it alone),
        written straight into one pinned buffer and sent by ONE non-blocking copy."""
        c = self.cfg
        shape = (n_chunks, len(seeds), c.num_frames_for_clip, c.dim_w)
        if os.environ.get("FLOAT_AMD_NOISE", "cpu").lower() == "device":  # per item what _noise_to_device draws for it alone
            return torch.cat([self._noise_to_device(n_chunks, sd) for sd in seeds], dim=1)
        buf = self.__dict__.get("_noise_pin_b")
        if buf is None or tuple(buf.shape) != shape:
            buf = self._noise_pin_b = torch.empty(shape, dtype=torch.float32, pin_memory=True)
        for i, sd in enumerate(seeds):
            g = torch.Generator("cpu")
            g.manual_seed(int(sd))
            for k in range(n_chunks):
                torch.randn(1, c.num_frames_for_clip, c.dim_w, generator=g, out=buf[k, i:i + 1])
        return buf.to(self.rank, non_blocking=True)

    def _noise_to_device(self, n_chunks, seed):
        """The reference's sequential CPU draws (fmt.draw_noise; FLOAT.py:203-215) written straight into a pinned buffer and sent
        by a non-blocking copy: the host neither waits for the encoder kernels queued in front of the copy nor leaves the GPU
        idle behind them.  The buffer is kept until the next clip (which starts after this one was synchronised)."""
        c = self.cfg
        shape = (n_chunks, 1, c.num_frames_for_clip, c.dim_w)
        if os.environ.get("FLOAT_AMD_NOISE", "cpu").lower() == "device":
            # "the reference on this device": torch.Generator(self.opt.rank) + randn(..., device=rank) per window (FLOAT.py:203-215) -
            # the stream a user of the reference on ROCm gets.  Default: the CPU generator's stream (the reference run on the CPU,
            # what the goldens and the oracle use); the two streams differ, the sampler does not.
            g = torch.Generator(self.rank)
            g.manual_seed(int(seed))
            return torch.stack([torch.randn(1, c.num_frames_for_clip, c.dim_w, device=self.rank, generator=g) for _ in range(n_chunks)])
        buf = self.__dict__.get("_noise_pin")
        if buf is None or tuple(buf.shape) != shape:
            buf = self._noise_pin = torch.empty(shape, dtype=torch.float32, pin_memory=True)
        g = torch.Generator("cpu")
        g.manual_seed(int(seed))
        for k in range(n_chunks):
            torch.randn(1, c.num_frames_for_clip, c.dim_w, generator=g, out=buf[k])
        return buf.to(self.rank, non_blocking=True)

    @torch.no_grad()
    def conditions_device(self, s, a, emo=None):
        """Once-per-clip stage on HIP operators, inputs already in HBM: s (1,3,H,W) in [-1,1] -> (s_r, feats handed to the
        decoder, r_s); a (N,) the normalised 16 kHz waveform -> (wa, T); and for any `emo` that is not one of the seven labels
        (None, 'none', 'S2E', ...) the speech-emotion scores (FLOAT.py:196-198)."""
        o = self.opt
        if host_models.emotion_index(emo) is None and self.emotion_predictor is None:
            raise NotImplementedError(
                "emotion='none' asks the speech-emotion model for scores (FLOAT.py:196-198) but the checkpoint has "
                "no `emotion_encoder.wav2vec2_for_emotion.` weights - pick an emotion or attach agent.emotion_predictor")
        T = math.ceil(a.shape[-1] * o.fps / o.sampling_rate)  # FLOAT.py:192
        need_ser = host_models.emotion_index(emo) is None
        # The speech-emotion model (emotion = "none", the reference's default widget) is independent of the other two producers
        # and the longest of the three (2.6 ms of ~200 small launches against 0.9 + 1.2): it runs on a side stream beside them
        # and joins before the FMT starts (111.8 -> 111.1 ms per clip, same box).  Only here: a second ACTIVE queue during the FMT chain
        # costs it 12 ms; and not for the label case - encoder || audio encoder on two streams measured
        # 0.45 ms SLOWER than one after the other (the stream hand-offs cost more than the 0.9 ms they hide).
        cur = torch.cuda.current_stream(self.rank)
        side = self._cond_stream() if (need_ser and os.environ.get("FLOAT_AMD_COND_STREAMS", "1") != "0") else None
        we = None
        if need_ser:
            if side is not None:
                side.wait_stream(cur)  # `a` was produced on the caller's stream
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                we = self.emotion_predictor(a).reshape(1, 1, -1).to(self.rank)
        wa = self.audio_encoder.inference(a, seq_len=T)
        s_r, _, _, r_s = self.enc.encode_image_into_latent(s, want_feats=False)  # FLOAT.py:283-291
        self.enc.hand_feats_to(self.G.dec)
        if side is not None:
            cur.wait_stream(side)
            we.record_stream(cur)  # allocated on the side stream, consumed on the caller's
        if we is None:
            we = self._one_hot(emo)
        return dict(s_r=s_r, feats=None, r_s=r_s, wa=wa, we=we, T=T)  # feats: already in the decoder (NHWC 16-bit)

    def _cond_stream(self):
        st = self.__dict__.get("_side_stream")
        if st is None:
            st = self._side_stream = torch.cuda.Stream(self.rank)
        return st

    def host_inputs(self, ref_img, ref_audio, no_crop=True):
        """Host plumbing of the reference's DataProcessor (generate.py:34-81): optional face crop, area resize to the model
        size, [-1,1]; mono, resample to 16 kHz, normalise.  Returns (s (1,3,H,W), a (N,)) on the device."""
        o = self.opt
        img = ref_img[0] if ref_img.dim() == 4 else ref_img
        if not no_crop:  # generate.py:77-78
            img, _ = host_models.process_img(img[..., :3].float(), o.input_size, getattr(o, "face_margin", 1.6), logger=main_logger)
        # The raw tensors cross PCIe first (3 MB + 0.6 MB for a 10-s clip) and the elementwise / reduction plumbing runs on the
        # device: on the host the same ops cost 1-29 ms per clip (torch's 128-thread intra-op pool on sub-megabyte tensors),
        # as much as a quarter of the whole clip.  The reference moves its slices to the device first too (nodes.py:193-201).
        s = host_models.preprocess_image(img[..., :3].to(self.rank, non_blocking=True), o.input_size)
        a = host_models.preprocess_audio(ref_audio["waveform"][0], ref_audio["sample_rate"], o.sampling_rate, device=self.rank)
        return s, a

    @torch.no_grad()
    def conditions(self, ref_img, ref_audio, emo=None, no_crop=True):
        s, a = self.host_inputs(ref_img, ref_audio, no_crop)
        return self.conditions_device(s, a, emo)

    @torch.no_grad()
    def infer_device(self, s, a, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, emo="S2E", seed=25, out=None):
        """Portrait and waveform in HBM -> (T,H,W,3) fp32 frames in [0,1] in pinned host memory: every operator of the path and
        the hand-over (frames of decode batch i leave inside the launches of batch i+1, float_dec_frames_host).  bench.py
        times exactly this call.  Like the reference, the grid size comes from opt.nfe (FLOAT.py:188)."""
        self.to_target()  # no-op while resident
        c = self.conditions_device(s, a, emo)  # encoder kernels enqueued; nothing below waits for them on the host
        n_chunks = int(math.ceil(c["T"] / self.cfg.num_frames_for_clip))
        noise = self._noise_to_device(n_chunks, seed if seed is not None else self.opt.seed)
        ov = os.environ.get("FLOAT_AMD_OVERLAP", "")  # "prio" | "cu:N": decode window k beside the chain of window k + 1 (pipeline.py)
        if ov and ov != "0":
            host = self.G.generate_to_host_overlap(c["r_s"], c["wa"], c["we"], c["s_r"], self.opt.nfe, a_cfg_scale, r_cfg_scale,
                                                   e_cfg_scale, noise=noise, out=out, mode=ov)
        else:
            host = self.G.generate_to_host(c["r_s"], c["wa"], c["we"], c["s_r"], None, self.opt.nfe, a_cfg_scale, r_cfg_scale,
                                           e_cfg_scale, noise=noise, out=out)
        torch.cuda.current_stream(self.rank).synchronize()  # the frames are in host memory
        self.G.release_host_inflight()
        if self.check_range("InferenceAgent.infer_device", allow_rebuild=True) == "rebuilt":
            return self.infer_device(s, a, a_cfg_scale, r_cfg_scale, e_cfg_scale, emo, seed, out)  # once more, in the wider types
        return host

    def range_counts(self, reset=True):
        """{operator: clamped / non-finite 16-bit stores since the last call} over every fp16 handle of the agent."""
        if self.G is None:
            return {}
        counts = self.G.range_counts(reset)
        for name, op in (("encoder", self.enc), ("audio", self.audio_encoder), ("speech_emotion", self.emotion_encoder)):
            if op is not None and op.dtype == "fp16":
                counts[name] = op.saturation(reset)
        return counts

    def check_range(self, where, allow_rebuild=False):
        """Once per clip, after the frames have arrived: an fp16 operator that left its range does not go unnoticed.
        FLOAT_AMD_RANGE = warn (default: RuntimeWarning + log line) | raise (Fp16RangeError) | off (skip the 8-byte reads) |
        auto: warn, then REBUILD the operators that overflowed in a type with the range - decoder + encoder in fp32 (the
        verification mode: same kernels, 1/16 of the MFMA rate), FMT / audio / speech-emotion in bf16 - and tell the caller to
        run the clip again ("rebuilt"); the agent keeps those types from then on."""
        mode = os.environ.get("FLOAT_AMD_RANGE", "warn").lower()
        if mode == "off":
            return {}
        bad = report_range(self.range_counts(), where, mode="warn" if mode == "auto" else mode)
        if mode == "auto" and bad and allow_rebuild:
            b, changed = self._build, False
            if ("decoder" in bad or "encoder" in bad) and b["dec_dtype"] != "fp32":
                b["dec_dtype"], changed = "fp32", True
            if "fmt" in bad and b["fmt_dtype"] == "fp16":
                b["fmt_dtype"], changed = "bf16", True
            if ("audio" in bad or "speech_emotion" in bad) and b["aud_dtype"] == "fp16":
                b["aud_dtype"], changed = "bf16", True
            if changed:
                main_logger.warning("%s: rebuilding the operators as fmt=%s, decoder/encoder=%s, audio=%s and running the clip again",
                                    where, b["fmt_dtype"], b["dec_dtype"], b["aud_dtype"])
                self.offload()
                self.to_target()
                return "rebuilt"
        return bad

    @torch.no_grad()
    def infer_device_batch(self, items, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, emo="S2E", seeds=None):
        """B clips of EQUAL length through one stacked FMT chain (float_fmt_sample_batch: every weight is read once per
        evaluation for all of them), then decoded one after the other.  items: [(s (1,3,H,W), a (N,))] in HBM; seeds: one per
        item (FloatProcess uses seed + i, nodes.py:189-209) - each item keeps its own noise stream, so item i is what
        infer_device gives for it alone (bit for bit where the GEMM tilings coincide, within the fp16 tolerance otherwise).
        Returns a list of (T,H,W,3) pinned host tensors."""
        self.to_target()
        B = len(items)
        seeds = list(seeds) if seeds is not None else [self.opt.seed] * B
        # once-per-clip producers of every item; the encoder's skip maps of item i are copied aside (33 MB, a D2D copy) so
        # that no item needs a second encoder pass when its turn to decode comes
        conds, feats = [], []
        slots = self.__dict__.setdefault("_feat_slots", [])
        for i, (s, a) in enumerate(items):
            conds.append(self.conditions_device(s, a, emo))
            if i >= len(slots):
                slots.append(None)
            slots[i] = self.enc.export_feats16(slots[i])
            feats.append(slots[i])
        T = conds[0]["T"]
        if any(c["T"] != T for c in conds):
            raise ValueError("infer_device_batch needs clips of equal length")
        n_chunks = int(math.ceil(T / self.cfg.num_frames_for_clip))
        noise = self._noise_batch_to_device(n_chunks, seeds)
        r_s = torch.cat([c["r_s"].reshape(1, -1) for c in conds])
        wa = torch.cat([c["wa"].reshape(1, T, -1) for c in conds])
        we = torch.cat([c["we"].reshape(1, 1, -1) for c in conds])
        r_d = self.G.batched_fmt(B).sample(r_s, wa, we, noise, self.opt.nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale)
        out = []
        for i in range(B):
            # decodes queue back to back: the last frames of item i cross PCIe inside the launches of item i + 1
            self.G.dec.set_feats16(feats[i], self.enc.dtype)
            out.append(self.G.decode_to_host(conds[i]["s_r"], r_d[i]))
        torch.cuda.current_stream(self.rank).synchronize()
        self.G.release_host_inflight()
        self.check_range("InferenceAgent.infer_device_batch")
        return out

    @torch.no_grad()
    def run_inference(self, res_video_path, ref_img, ref_audio, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0,
                      emo="S2E", nfe=10, no_crop=False, seed=25):
        """Reference signature (generate.py:154-173).  Returns (T,H,W,3) fp32 in [0,1] on the CPU (pinned).
        Like the reference, the grid size comes from opt.nfe, not from the `nfe` argument (FLOAT.py:188)."""
        s, a = self.host_inputs(ref_img, ref_audio, no_crop)
        return self.infer_device(s, a, a_cfg_scale, r_cfg_scale, e_cfg_scale, emo, seed)
