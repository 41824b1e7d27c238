"""What does residency cost?  InferenceAgent build / offload() / to_target() at the checkpoint shapes (512 px, wav2vec2-base audio
encoder, wav2vec2-large speech-emotion model), per operator - the number behind INTEGRATION.md's model_to_target paragraph."""
import importlib, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
dev = torch.device("cuda:0")
torch.cuda.init()
cfg = pkg.config.FmtConfig()
gen = importlib.import_module(pkg.__name__ + ".src.nodes.generate")
opt = importlib.import_module(pkg.__name__ + ".src.nodes.options.base_options").BaseOptions()
opt.nfe, opt.input_size, opt.fps, opt.rank = 11, 512, 25.0, dev
acfg, ecfg = pkg.config.AudioConfig(), pkg.config.emotion_audio_config()
t0 = time.perf_counter()
parts = dict(enc=pkg.weights.synth_encoder_state(512, seed=1), dec=pkg.weights.synth_decoder_state(512, seed=1),
             fmt=pkg.weights.synth_fmt_state(cfg, seed=1), audio_encoder=(pkg.weights.synth_audio_state(acfg, seed=1), acfg),
             emotion_encoder=(pkg.weights.synth_audio_state(ecfg, seed=2), ecfg))
print("synthesising the host weights: %.2f s" % (time.perf_counter() - t0))


def timed(what, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("%-44s %8.1f ms" % (what, (time.perf_counter() - t) * 1e3), flush=True)
    return r


agent = timed("InferenceAgent(...) first build", lambda: gen.InferenceAgent(opt, parts, dev, max_frames=32))
img = torch.rand(1, 3, 512, 512, device=dev) * 2 - 1
wav = pkg.weights.synth_waveform(2.0, seed=1).to(dev)
first = timed("first clip (2 s, nfe 11)", lambda: agent.infer_device(img, wav, 2.0, 1.0, 1.0, emo="neutral", seed=3)).clone()
timed("second clip", lambda: agent.infer_device(img, wav, 2.0, 1.0, 1.0, emo="neutral", seed=3))
for rep in range(2):
    free0 = torch.cuda.mem_get_info()[0]
    timed("offload()", agent.offload)
    print("   HBM freed: %.2f GB" % ((torch.cuda.mem_get_info()[0] - free0) / 2**30))
    timed("to_target()", agent.to_target)
    again = timed("clip after to_target()", lambda: agent.infer_device(img, wav, 2.0, 1.0, 1.0, emo="neutral", seed=3))
    print("   bitwise the first clip:", torch.equal(again, first))
# per operator
timed("offload()", agent.offload)
b = agent._build
timed("  FMT handle", lambda: pkg.fmt.FlowMatchingTransformerHIP(parts["fmt"], cfg, dev, b["fmt_dtype"], b["use_graph"], 1))
timed("  decoder handle", lambda: pkg.decoder.SynthesisHIP(parts["dec"], 512, 512, dev, b["dec_dtype"], 32))
timed("  encoder handle", lambda: pkg.encoder.EncoderHIP(parts["enc"], 512, 512, 20, dev, dtype=b["dec_dtype"], direction_weight=parts["dec"]["direction.weight"]))
timed("  audio encoder handle", lambda: pkg.audio.AudioEncoderHIP(parts["audio_encoder"][0], acfg, dev, dtype=b["aud_dtype"]))
timed("  speech-emotion handle", lambda: pkg.audio.Audio2EmotionHIP(parts["emotion_encoder"][0], ecfg, dev, dtype=b["aud_dtype"]))
