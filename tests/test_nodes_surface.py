"""Drop-in check of the node surface against the contract captured from the reference classes
(tests/golden/node_surface.json, written by tools/make_goldens.py): widget dictionaries, return
tuples, names, BaseOptions defaults and the batch/seed schedule of FloatProcess.floatprocess."""
import json
import os

import torch

from tests.util import GOLDEN, load_pkg

pkg = load_pkg()
nodes = pkg.NODE_CLASS_MAPPINGS
FIX = json.load(open(os.path.join(GOLDEN, "node_surface.json")))


def _norm(x):
    return json.loads(json.dumps(x, default=str))


def test_node_contracts():
    for name in ("FloatProcessOpt", "FloatAdvancedParameters", "LoadFloatModelsOpt"):
        cls, ref = nodes[name], FIX[name]
        for attr in ("RETURN_TYPES", "RETURN_NAMES"):
            assert list(getattr(cls, attr)) == ref[attr], (name, attr)
        for attr in ("FUNCTION", "CATEGORY", "UNIQUE_NAME", "DISPLAY_NAME"):
            assert getattr(cls, attr) == ref[attr], (name, attr)
        it, rit = _norm(cls.INPUT_TYPES()), ref["INPUT_TYPES"]
        assert list(it["required"].keys()) == list(rit["required"].keys()), name
        if name != "LoadFloatModelsOpt":
            assert it == rit, name
        else:  # model / device lists depend on the host; everything else must match
            assert it["optional"] == rit["optional"]
            assert it["required"]["cudnn_benchmark"] == rit["required"]["cudnn_benchmark"]
            assert it["required"]["model"][1] == rit["required"]["model"][1]
    assert pkg.NODE_DISPLAY_NAME_MAPPINGS["FloatProcessOpt"] == "FLOAT Process (Opt)"


def test_base_options_defaults():
    from dataclasses import asdict
    BaseOptions = pkg.src.nodes.options.base_options.BaseOptions if hasattr(pkg, "src") else None
    import importlib
    BaseOptions = importlib.import_module(pkg.__name__ + ".src.nodes.options.base_options").BaseOptions
    mine = {k: v for k, v in asdict(BaseOptions()).items()}
    for k, v in FIX["base_options"].items():
        assert k in mine and str(mine[k]) == str(v), k
    adv = nodes["FloatAdvancedParameters"]().get_options(1.0, 2, 0.1, 0.1, 0.1, 1e-5, 1e-5, 10, "euler", 1.6,
                                                         "blend_with_color", "#000000")[0]
    assert list(adv.keys()) == list(FIX["FloatAdvancedParameters"]["INPUT_TYPES"]["required"].keys())
    assert all(hasattr(BaseOptions(), k) for k in adv)


def test_floatprocess_batch_schedule():
    calls = []

    class FakePipe:
        class opt:
            r_cfg_scale = 1.0
            fps = 25.0

        def run_inference(self, path, img, audio, **kw):
            calls.append(dict(image_mark=float(img[0, 0, 0, 0]), audio_mark=float(audio["waveform"][0, 0, 0]), seed=kw["seed"],
                              emo=kw["emo"], no_crop=kw["no_crop"], a=kw["a_cfg_scale"], e=kw["e_cfg_scale"], r=kw["r_cfg_scale"]))
            return torch.full((2, 4, 4, 3), float(len(calls)))

    img = torch.zeros(2, 4, 4, 3)
    img[0] += 10
    img[1] += 11
    wav = torch.zeros(3, 1, 8)
    for i in range(3):
        wav[i] += 20 + i
    out = nodes["FloatProcessOpt"]().floatprocess(img, {"waveform": wav, "sample_rate": 16000}, FakePipe(), 2.0, 1.0, 30.0,
                                                  "happy", False, 1000)
    s = FIX["schedule"]
    assert calls == s["calls"]
    assert list(out[0].shape) == s["images_shape"]
    assert [float(out[0][i, 0, 0, 0]) for i in range(out[0].shape[0])] == s["images_first"]
    assert list(out[1]["waveform"].shape) == s["audio_shape"] and out[1]["waveform"][0, 0].tolist() == s["audio_values"]
    assert out[2] == s["fps"] and FakePipe.opt.fps == 30.0

    # the stacked-chain path (InferenceAgent.infer_device_batch) keeps the reference's schedule: item i = image min(i, Bi-1),
    # audio min(i, Ba-1), seed + i, same scales / emotion / crop flag, and the same outputs
    calls2 = []

    class BatchPipe(FakePipe):
        def host_inputs(self, img, audio, no_crop):
            return (float(img[0, 0, 0, 0]), float(audio["waveform"][0, 0, 0]), no_crop)

        def infer_device_batch(self, items, a, r, e, emo, seeds):
            for (im, au, nc), sd in zip(items, seeds):
                calls2.append(dict(image_mark=im, audio_mark=au, seed=sd, emo=emo, no_crop=nc, a=a, e=e, r=r))
            return [torch.full((2, 4, 4, 3), float(i + 1)) for i in range(len(items))]

    out2 = nodes["FloatProcessOpt"]().floatprocess(img, {"waveform": wav, "sample_rate": 16000}, BatchPipe(), 2.0, 1.0, 30.0,
                                                   "happy", False, 1000)
    assert calls2 == s["calls"] and torch.equal(out2[0], out[0]) and torch.equal(out2[1]["waveform"], out[1]["waveform"])


def test_host_preprocessing():
    hm = pkg.host_models
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(2, 16000, generator=g) * 0.1 + 0.3
    a = hm.preprocess_audio(wav, 16000)
    assert a.shape == (1, 16000) and abs(float(a.mean())) < 1e-5 and abs(float(a.std(unbiased=False)) - 1) < 1e-3
    img = torch.rand(600, 600, 3, generator=g)
    x = hm.preprocess_image(img, 512)
    assert x.shape == (1, 3, 512, 512) and float(x.min()) >= -1 and float(x.max()) <= 1
    assert hm.emotion_one_hot("happy").tolist() == [[[0, 0, 0, 1, 0, 0, 0]]]


def test_unified_checkpoint_split():
    gen = importlib_generate()
    state = {"fmt.blocks.0.attn.qkv.weight": 1, "motion_autoencoder.dec.conv1.conv.weight": 2,
             "motion_autoencoder.enc.fc.0.weight": 3, "audio_encoder.audio_projection.0.weight": 4,
             "audio_encoder.wav2vec2.encoder.layers.0.x": 5, "emotion_encoder.wav2vec2_for_emotion.classifier.dense.weight": 6}
    parts = gen.split_unified(state)
    assert parts["fmt"] == {"blocks.0.attn.qkv.weight": 1} and parts["dec"] == {"conv1.conv.weight": 2}
    assert parts["enc"] == {"fc.0.weight": 3} and parts["proj"] == {"0.weight": 4}
    assert list(parts["wav2vec"]) == ["encoder.layers.0.x"] and list(parts["ser"]) == ["classifier.dense.weight"]


def importlib_generate():
    import importlib
    return importlib.import_module(pkg.__name__ + ".src.nodes.generate")


VA = json.load(open(os.path.join(GOLDEN, "node_surface_va.json")))
VA_BUILT = ("LoadFloatEncoderModel", "LoadFloatSynthesisModel", "LoadFMTModel", "LoadWav2VecModel", "LoadAudioProjectionLayer",
            "ApplyFloatEncoder", "FloatGetIdentityReferenceVA", "FloatSampleMotionSequenceRD_VA", "ApplyFloatSynthesis",
            "FloatAudioPreprocessAndFeatureExtract", "FloatApplyAudioProjection", "LoadEmotionRecognitionModel",
            "FloatExtractEmotionWithCustomModel", "FloatExtractEmotionWithCustomModelDyn")


def test_va_node_contracts():
    """Very-advanced loaders / stage nodes: same widgets (name, order, type, default, range), return tuples and names as
    the reference classes (nodes_vadv_loader.py, nodes_vadv.py); file / device lists depend on the host."""
    for name in VA_BUILT:
        cls, ref = nodes[name], VA[name]
        for attr in ("RETURN_TYPES", "RETURN_NAMES"):
            assert list(getattr(cls, attr)) == ref[attr], (name, attr)
        for attr in ("FUNCTION", "CATEGORY", "UNIQUE_NAME", "DISPLAY_NAME"):
            assert getattr(cls, attr) == ref[attr], (name, attr)
        it, rit = _norm(cls.INPUT_TYPES()), ref["INPUT_TYPES"]
        assert list(it["required"].keys()) == list(rit["required"].keys()), name
        assert it.get("optional", {}) == rit.get("optional", {}), name
        for k, v in rit["required"].items():
            if k == "target_device" or k.endswith("_file") or k == "model_folder":
                assert isinstance(it["required"][k][0], list) and it["required"][k][0], (name, k)
                if k != "target_device":
                    assert it["required"][k][0] == v[0], (name, k)  # default file name when the folder is empty
            else:
                assert it["required"][k] == v, (name, k)
        assert pkg.NODE_DISPLAY_NAME_MAPPINGS[name] == ref["DISPLAY_NAME"] + " " + ref["SUFFIX"]
    assert set(VA) == set(VA_BUILT)  # every class of the reference's two VA modules is there


def test_va_part_extraction(tmp_path, monkeypatch):
    """utils/downloader.py:35-42 layout: a part that is missing on disk is cut out of the unified FLOAT.safetensors with
    its prefix stripped; with neither file the loader raises FileNotFoundError."""
    import importlib
    from safetensors.torch import load_file, save_file
    L = importlib.import_module(pkg.__name__ + ".src.nodes.nodes_vadv_loader")
    monkeypatch.setenv("FLOAT_MODELS_DIR", str(tmp_path))
    import pytest
    with pytest.raises(FileNotFoundError):
        L.ensure_model_part_exists("fmt", L.FMT_SUBDIR, "fmt.safetensors")
    os.makedirs(tmp_path / "float")
    uni = {"fmt.x_embedder.proj.weight": torch.ones(4, 2), "motion_autoencoder.enc.fc.0.bias": torch.zeros(3),
           "audio_encoder.audio_projection.0.weight": torch.full((2, 2), 2.0)}
    save_file(uni, str(tmp_path / "float" / "FLOAT.safetensors"))
    p = L.ensure_model_part_exists("fmt", L.FMT_SUBDIR, "fmt.safetensors")
    assert p == str(tmp_path / "float" / "fmt" / "fmt.safetensors") and list(load_file(p)) == ["x_embedder.proj.weight"]
    p = L.ensure_model_part_exists("projection", L.AUDIO_PROJ_DIR, "projection.safetensors")
    assert list(load_file(p)) == ["0.weight"]
    assert L.safe_parse_list_str("[1, 3, 3, 1]") == [1, 3, 3, 1]
    with pytest.raises(ValueError):
        L.safe_parse_list_str("__import__('os')")
