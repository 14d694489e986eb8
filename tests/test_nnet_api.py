"""CPU: the host-side mirror keeps the reference's API surface -- state_dict keys/shapes, parameter count, seeded initialisation
(bit-identical to the reference under the same torch seed), registries, length arithmetic -- and the C ABI is complete."""
import os
import subprocess
import sys

import pytest
import torch

from tests.helpers import load_json

TORCHAUDIO_KEYS = {"encoder.audio_encoder.audio_preprocessing.Spectrogram.window", "encoder.audio_encoder.audio_preprocessing.MelScale.fb"}


@pytest.fixture(scope="module")
def av_model():
    import nnet
    torch.manual_seed(0)
    return nnet.AudioVisualEfficientConformerInterCTC()


def test_state_dict_matches_reference(av_model):
    g = load_json("av_full_seed0")
    sd = av_model.state_dict()
    ref = {k: (tuple(s), d) for k, s, d in g["state_dict"]}
    mine = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()}
    assert set(mine) - set(ref) == TORCHAUDIO_KEYS            # real torchaudio registers these two buffers persistently
    assert set(ref) - set(mine) == set()
    for k, v in ref.items():
        assert mine[k] == v, k
    assert [k for k, _ in av_model.named_parameters()] == g["param_names"]
    assert sum(p.numel() for p in av_model.parameters()) == g["n_params"] == 61738836


def test_seeded_init_is_bit_identical_to_reference(av_model):
    g = load_json("av_full_seed0")
    sd = av_model.state_dict()
    for k, (s, a) in g["param_checksums"].items():
        v = sd[k].double()
        # (sums are accumulated in storage order, which differs for channels-last weights: compare to 1e-12)
        assert abs(float(v.sum()) - s) <= 1e-12 * max(1.0, a) and abs(float(v.abs().sum()) - a) <= 1e-12 * max(1.0, a), k


def test_weight_storage_layouts(av_model):
    enc = av_model.encoder
    w = enc.video_encoder.front_end[3].blocks[0].layers[0].weight
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)            # physical [Cout][KH][KW][Cin]
    dw = enc.audio_encoder.back_end.conformer_blocks[0].conv_module.layers[3].weight
    assert dw.shape == (180, 1, 15) and dw.stride()[0] == 1 and dw.stride()[2] == 180  # physical [K][C]


def test_registries_and_names():
    import nnet
    for name in ["RelPos1dMultiHeadAttention", "RelPosPatch1dMultiHeadAttention"]:
        assert name in nnet.attentions.att_dict
    for name in ["Linear", "Conv1d", "Conv2d", "Conv3d", "MaxPool3d", "Dropout"]:
        assert name in nnet.layers.layer_dict
    for name in ["LayerNorm", "BatchNorm1d", "BatchNorm2d", "BatchNorm3d"]:
        assert name in nnet.normalizations.norm_dict
    assert "ConformerBlock" in nnet.blocks.block_dict and "Swish" in nnet.activations.act_dict and "he_normal" in nnet.initializations.init_dict
    for cls in ["AudioVisualEfficientConformerInterCTC", "AudioEfficientConformerInterCTC", "VisualEfficientConformerInterCTC", "CTCLoss", "Adam",
                "NoamDecayScheduler", "CTCGreedySearchDecoder", "CTCBeamSearchDecoder", "WordErrorRate", "CollateFn", "Permute", "TimeMaskSecond", "Model", "Module", "Mask"]:
        assert hasattr(nnet, cls), cls
    assert hasattr(nnet.datasets, "LRS") and hasattr(nnet.datasets, "MultiDataset")


def test_other_models_param_counts():
    import nnet
    assert sum(p.numel() for p in nnet.AudioEfficientConformerInterCTC(interctc_blocks=[]).parameters()) == load_json("ao_cfg1_seed0")["n_params"]
    assert sum(p.numel() for p in nnet.VisualEfficientConformerInterCTC().parameters()) == 40903112


def test_no_cpu_fallback(av_model):
    """the product path must fail loudly without a GPU (the CPU restatement lives in oracle/ only)"""
    import nnet
    lin = nnet.Linear(8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lin(torch.randn(2, 8))
    with pytest.raises(RuntimeError):
        av_model([torch.randn(1, 4, 88, 88, 1), torch.tensor([4]), torch.randn(1, 2560), torch.tensor([2560])])


def test_scheduler_and_mask_index_work():
    import nnet
    j = load_json("int_cases")
    sch = nnet.NoamDecayScheduler(10000, 360, 2)
    for s, v in j["noam_lr"].items():
        assert abs(sch.get_val_step(int(s)) - v) <= 1e-15 * max(1.0, abs(v)) + 1e-18
    m = nnet.Mask()(torch.zeros(3, 8, 4), torch.tensor([8, 5, 1]))
    assert m.tolist() == j["mask_T8"]
    from avec_amd.ops import _pool_mask
    assert _pool_mask(m, 8, 3).unsqueeze(1).tolist() == j["patch_mask_T8_P3"]
    from avec_amd.nnet.modules import LengthMask
    lm = LengthMask(torch.tensor([8, 5, 1])).strided(2)
    assert lm.lengths.tolist() == [4, 3, 1]
    # a LengthMask is the same mask as the reference's strided slicing mask[:, :, ::2, ::2]
    assert nnet.Mask()(torch.zeros(3, 4, 4), lm.lengths).tolist() == m[:, :, ::2, ::2].tolist()
    from avec_amd.nnet.decoders import ctc_collapse
    assert ctc_collapse([0, 1, 1, 0, 1, 2, 2, 2, 0, 0, 3, 4, 4], 13) == [1, 1, 2, 3, 4]


def test_c_abi_exports_every_declared_symbol():
    from avec_amd.lib import LIB_PATH, declared_functions, lib
    decl = declared_functions()
    assert len(decl) >= 40
    out = subprocess.run("nm -D --defined-only %s" % LIB_PATH, shell=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(decl) <= exported, sorted(set(decl) - exported)
    lib.load()                                   # resolves + types every symbol; raises if one is missing
    assert lib.raw("avec_version")() == 4
    from avec_amd.lib import ABI_STRUCTS
    import ctypes
    for which, st in enumerate(ABI_STRUCTS):
        assert lib.raw("avec_struct_size")(which) == ctypes.sizeof(st), st.__name__
    assert lib.raw("avec_ctc_workspace_floats")(2, 10, 3) == 2 * (10 * 7 + 10)


def test_c_abi_argument_validation_without_gpu():
    """argument errors are reported before any launch (no GPU needed)"""
    from avec_amd.lib import lib
    with pytest.raises(RuntimeError, match="layernorm_fwd"):
        lib.layernorm_fwd(0, None, None, None, None, 0, None, None, 4, 8, 1e-6, None)
    with pytest.raises(RuntimeError, match="bad dims"):
        lib.ctc_loss(1, 1, 1, 1, 1, None, None, 1, 0, 5, 4, 2, 0, 1, None)


def test_lrw_classifier_api_and_seeded_init_match_reference():
    """BASELINE config 1 (VisualEfficientConformerCE, nnet/models_zoo.py:33-62): state_dict keys/shapes, parameter count and the seed-0
    initial values equal the reference's (tests/golden/lrw_ce_seed0.json, written by tests/golden/make_golden.py lrw)."""
    import nnet
    g = load_json("lrw_ce_seed0")
    torch.manual_seed(0)
    model = nnet.VisualEfficientConformerCE(vocab_size=500)
    model.compile()
    sd = model.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == g["state_dict"]
    assert sum(p.numel() for p in model.parameters()) == g["n_params"] == 40489740
    for k, (s1, s2) in g["param_checksums"].items():
        v = sd[k].double()
        assert abs(float(v.sum()) - s1) <= 1e-6 * max(1.0, abs(s1)) and abs(float(v.abs().sum()) - s2) <= 1e-6 * max(1.0, abs(s2)), k
    assert nnet.CategoricalAccuracy()(torch.tensor([1, 2, -1]), torch.tensor([[0., 1., 0.], [0., 1., 0.], [1., 0., 0.]])) == 50.0


def test_grouped_audio_encoder_state_dict_and_init_match_reference():
    """AudioEfficientConformerEncoder(att_type="grouped") (nnet/networks.py:389-392): same state_dict keys / shapes, same parameter count and the same seeded
    initialisation (scaled_uniform weights, zero biases, zero u / v) as the reference (tests/golden/ao_grouped_state.json)"""
    import json
    import os
    import nnet
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ao_grouped_state.json")))
    torch.manual_seed(0)
    enc = nnet.AudioEfficientConformerEncoder(att_type="grouped", interctc_blocks=[])
    sd = enc.state_dict()
    ref_keys = [k for k in g["keys"] if not k.endswith(("Spectrogram.window", "MelScale.fb"))]
    mine = [k for k in sd.keys() if not k.endswith(("Spectrogram.window", "MelScale.fb"))]
    assert mine == ref_keys
    for k, shp in zip(g["keys"], g["shapes"]):
        if k in sd:
            assert list(sd[k].shape) == shp, k
    assert sum(p.numel() for p in enc.parameters()) == g["n_params"]
    for k, v in g["probe"].items():
        if k in sd:
            assert abs(float(sd[k].double().sum()) - v) <= 1e-6 * max(1.0, abs(v)), k


def test_word_error_rate_standardizes_and_is_corpus_level():
    """nnet/metrics.py:101-110: 100 * jiwer.wer(targets, outputs, standardize=True) -- lower-casing, contraction expansion, Kaldi non-words and white space are
    normalised before counting, and the rate is errors over ALL reference words of the list (not the mean of per-sentence rates)."""
    from avec_amd.nnet.metrics import WordErrorRate, standardize
    assert standardize("I  CAN'T  [noise] go <unk> It's") == "i can not go it is"
    wer = WordErrorRate()
    assert wer(["It's a test"], ["it is a TEST"]) == 0.0
    # 1 substitution in a 2-word sentence + 0 errors in an 8-word sentence: corpus level 10 %, mean of sentence rates 25 %
    assert abs(wer(["hello world", "a b c d e f g h"], ["hello word", "a b c d e f g h"]) - 10.0) < 1e-9
    assert abs(wer(["a b c"], ["a c"]) - 100.0 / 3) < 1e-9 and abs(wer(["a b"], ["a x b y"]) - 100.0) < 1e-9      # deletion; two insertions


def test_integration_md_stub_matches_the_header():
    """INTEGRATION.md's generated struct stubs (tools/gen_integration.py) are current, and what a maintainer would paste has the library's own struct sizes"""
    import ctypes
    import re
    from avec_amd.lib import Epilogue, Rows, declared_functions, lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_integration.py"), "--check"]).returncode == 0, "INTEGRATION.md is stale: python tools/gen_integration.py"
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    blk = doc[doc.index("<!-- BEGIN GENERATED"):doc.index("<!-- END GENERATED -->")]
    ns = {"ctypes": ctypes}
    exec(blk[blk.index("```python") + 9:blk.rindex("```")], ns)
    assert ctypes.sizeof(ns["Epilogue"]) == ctypes.sizeof(Epilogue) == lib.raw("avec_struct_size")(1)
    assert ctypes.sizeof(ns["Rows"]) == ctypes.sizeof(Rows) == lib.raw("avec_struct_size")(0)
    assert [f[0] for f in ns["Epilogue"]._fields_] == [f[0] for f in Epilogue._fields_]
    assert int(re.search(r"\((\d+) `extern \"C\"` symbols", doc).group(1)) == len(declared_functions())
