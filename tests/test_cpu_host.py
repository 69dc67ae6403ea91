"""CPU tier: host-side logic of the product (no GPU compute): the C-ABI library loads and exports every symbol
include/eoe_hip.h declares, argument validation returns errors (not crashes), the step-batch layout and the
metrics match the oracle / golden vectors bit-exactly where they are integer paths."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import batching, fill, metrics as ometrics


def test_library_exports_every_declared_symbol():
    from eoe_amd import _lib
    declared = _lib.header_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(_lib.lib, name), f"libeoe_hip.so does not export {name}"
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    assert _lib.lib.eoe_abi_version() == _lib.ABI_VERSION
    # struct layouts the Python side mirrors
    assert C.sizeof(_lib.AdamChunk) == 40 and C.sizeof(_lib.AdamScalars) == 36 and C.sizeof(_lib.AdamTile) == 64
    mirrors = [_lib.GemmArgs, _lib.ConvGeometry, _lib.AdamChunk, _lib.AdamScalars, _lib.VitBlockFwdArgs, _lib.VitBlockBwdArgs,
               _lib.CGateArgs, _lib.CGateBwdArgs, _lib.SGateArgs, _lib.SGateBwdArgs, _lib.AdamTile, _lib.RedTable]
    for which, cls in enumerate(mirrors):
        assert _lib.lib.eoe_struct_size(which) == C.sizeof(cls), (cls.__name__, _lib.lib.eoe_struct_size(which), C.sizeof(cls))
    assert _lib.lib.eoe_struct_size(len(mirrors)) == -1


def test_argument_errors_are_reported_not_fatal():
    from eoe_amd import _lib
    g = _lib.GemmArgs()                 # all-null arguments
    rc = _lib.lib.eoe_gemm_nt(C.byref(g), None)
    assert rc == 1 and b"null" in _lib.lib.eoe_last_error()
    rc = _lib.lib.eoe_attn_fwd(None, None, 1, 50, 12, 1, None)
    assert rc == 1
    with pytest.raises(_lib.EoeError):
        _lib.check(_lib.lib.eoe_layernorm_fwd(None, 0, None, None, None, None, 1, 768, 1e-5, 1, 0, None), "ln")


def test_ops_refuse_cpu_tensors():
    import eoe_amd
    with pytest.raises(RuntimeError):
        eoe_amd.hsc_loss(torch.zeros(2, 4), torch.zeros(2, dtype=torch.long))
    from eoe_amd.models import ClipViTB32Custom
    m = ClipViTB32Custom(layers=1)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 224, 224))


def test_batch_layout_matches_oracle_bit_exact():
    from eoe_amd import data, parallel
    rng = np.random.default_rng(0)
    for n_normal_ds, bs, oe_sizes in ((100, 7, [4, 4, 4]), (33, 16, [16, 16]), (10, 5, [2, 2, 2, 2])):
        normal = (rng.standard_normal((bs, 2)).astype(np.float32), np.zeros(bs, np.int64), rng.integers(0, n_normal_ds, bs))
        chunks = [(rng.standard_normal((k, 2)).astype(np.float32), np.ones(k, np.int64), rng.integers(0, 50, k)) for k in oe_sizes]
        want = batching.balanced_concat(normal, iter(chunks), n_normal_ds)
        got = data.balanced_concat([torch.from_numpy(a) for a in normal],
                                   iter([[torch.from_numpy(a) for a in c] for c in chunks]), n_normal_ds)
        for w, g in zip(want, got):
            assert np.array_equal(w, g.numpy())
    assert data.tile_oe_indices(torch.tensor([4, 9]), 5).tolist() == batching.tile_oe_indices(np.array([4, 9]), 5).tolist()
    for nn, no, world in ((128, 128, 8), (5, 5, 2), (7, 3, 4), (1, 1, 2)):
        for r in range(world):
            assert parallel.shard_rows(nn, no, r, world).tolist() == batching.shard_rows(nn, no, r, world).tolist()


def test_synthetic_source_honours_the_contract():
    from eoe_amd.data import SyntheticAD
    ds = SyntheticAD(n_train_normal=20, n_oe=6, n_test=10, res=8, seed=3)
    train, test = ds.loaders(8)
    batches = list(train)
    assert len(batches) == len(train) == 3
    for imgs, lbls, idcs in batches[:2]:
        assert lbls.tolist() == [0] * 8 + [1] * 8 and imgs.shape == (16, 3, 8, 8)
        assert (idcs[:8] < 20).all() and (idcs[8:] >= 20).all() and (idcs[8:] < 26).all()     # OE offset rule
    imgs, lbls, idcs = batches[2]                                                              # ragged last batch
    assert lbls.tolist() == [0] * 4 + [1] * 4
    assert sorted(torch.cat([b[2][: len(b[2]) // 2] for b in batches]).tolist()) == list(range(20))
    assert sum(len(b[1]) for b in test) == 10


def test_metrics_match_golden(golden):
    from eoe_amd import metrics
    g = golden("g7_metrics")
    for i in range(4):
        n, ties = int(g[f"n{i}"]), bool(g[f"ties{i}"])
        s = fill.fill(f"g7/s{i}", (n,), std=1.0)
        if ties:
            s = np.round(s * 4) / 4
        y = fill.fill_int(f"g7/y{i}", (n,), 0, 2)
        y[0], y[1] = 0, 1
        assert abs(metrics.roc_auc(y, s) - float(g[f"auc{i}"])) < 1e-12
        assert abs(metrics.average_precision(y, s) - float(g[f"ap{i}"])) < 1e-12
        assert abs(metrics.roc_auc(y, s) - ometrics.roc_auc(y, s)) < 1e-15
    assert np.isnan(metrics.roc_auc(np.ones(3), np.arange(3.0)))


def test_state_dict_names_match_reference_names():
    """parameter names/shapes of the drop-in modules equal the oracle's (= the reference's)"""
    from eoe_amd.models import ClipViTB32Custom
    from oracle import models as omodels
    a = {k: tuple(v.shape) for k, v in ClipViTB32Custom(layers=2).state_dict().items()}
    b = {k: tuple(v.shape) for k, v in omodels.ClipViTNet(layers=2).state_dict().items()}
    assert a == b
    a = {k: tuple(v.shape) for k, v in ClipViTB32Custom(layers=1, clf=True).state_dict().items()}
    assert a["final_linear.weight"] == (1, 512)
    m = ClipViTB32Custom(layers=1, freeze=True)
    assert m.freeze_parts() and all(not p.requires_grad for p in m.feature_model.parameters())
    assert all(p.requires_grad for p in m.final_linear.parameters())


def test_trainer_registry_and_hooks():
    from eoe_amd.training import TRAINER, ADTrainer
    assert set(TRAINER) == {"hsc", "bce", "dsvdd", "dsad", "focal", "clip"}  # training/__init__.py:8-11
    for cls in TRAINER.values():
        assert issubclass(cls, ADTrainer)
        for hook in ("prepare_metric", "compute_anomaly_score", "loss", "train_cls", "eval_cls", "run", "load"):
            assert callable(getattr(cls, hook))


def test_header_is_plain_c():
    """the drop-in boundary must be consumable from C (cgo / JNI / ctypes generators): include/eoe_hip.h parses as C99"""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "eoe_hip.h")
    r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_resize_coefficient_tables_equal_the_oracle():
    """the library's HOST helper eoe_resize_coeffs (Pillow's precompute_coeffs + 8 bpc normalisation) against the oracle's table,
    which the fixture g14 pins to Pillow: identical integers for down- and up-scaling, both filters"""
    import ctypes as C
    import numpy as np
    from eoe_amd import _lib
    from oracle import augment
    for in_size, out_size in ((100, 48), (75, 150), (256, 224), (500, 256), (32, 32), (7, 3)):
        for name, code in (("bilinear", _lib.EOE_RESIZE_BILINEAR), ("bicubic", _lib.EOE_RESIZE_BICUBIC)):
            k = C.c_int(0)
            _lib.check(_lib.lib.eoe_resize_coeffs(in_size, out_size, code, None, None, 0, C.byref(k)), "size query")
            bounds = np.zeros((out_size, 2), np.int32)
            kk = np.full((out_size, k.value), -7, np.int32)
            _lib.check(_lib.lib.eoe_resize_coeffs(in_size, out_size, code, bounds.ctypes.data, kk.ctypes.data, k.value, None), "tables")
            rb, rk = augment.resize_coeffs(in_size, out_size, name)
            assert rk.shape[1] == k.value and np.array_equal(bounds, rb), (in_size, out_size, name)
            assert np.array_equal(kk, rk), (in_size, out_size, name)
    assert _lib.lib.eoe_resize_coeffs(0, 4, _lib.EOE_RESIZE_BILINEAR, None, None, 0, None) != 0


def test_grad_scale_api():
    """the fp16 gradient scale is a power of two, 256 by default for fp16 compute and 1 for bf16; the 1x1-map shortcut only fires on
    its own geometry (host logic, no GPU)"""
    import torch
    import eoe_amd
    from eoe_amd import ops
    assert eoe_amd.grad_scale() == 1.0
    assert eoe_amd.default_grad_scale(torch.float16) == 256.0 and eoe_amd.default_grad_scale(torch.bfloat16) == 1.0
    eoe_amd.set_grad_scale(4096)
    assert eoe_amd.grad_scale() == 4096.0
    eoe_amd.set_grad_scale(1.0)
    for bad in (0, -2, 3, 0.3):
        with pytest.raises(ValueError):
            eoe_amd.set_grad_scale(bad)
    # (n, H, W, C, kh, kw, stride, pad, Ho, Wo)
    assert ops._single_pixel((256, 1, 1, 512, 3, 3, 1, 1, 1, 1))
    assert not ops._single_pixel((256, 2, 2, 512, 3, 3, 1, 1, 2, 2))
    assert not ops._single_pixel((256, 1, 1, 512, 3, 3, 2, 1, 1, 1))
    assert not ops._single_pixel((256, 1, 1, 48, 3, 3, 1, 1, 1, 1))


def test_task_definition_matches_oracle_and_golden(golden):
    """host logic of the class x seed loop (ADTrainer.get_nominal_classes, data.normal_subset / ad_targets) against the oracle and
    the fixture made by the reference's own functions (g15): bit-exact"""
    from eoe_amd import data
    from eoe_amd.training import TRAINER
    g = golden("g15_tasks")
    for n in (3, 10, 30):
        for mode in ("one_vs_rest", "leave_one_out", "fifty_fifty"):
            tr = TRAINER["hsc"](None, dataset=None, classes=[str(i) for i in range(n)], ad_mode=mode)
            for c in range(n):
                got = tr.get_nominal_classes(c)
                assert got == batching.nominal_classes(mode, c, n) == g[f"nominal/{n}/{mode}/{c}"].tolist()
    for name in ("small", "cifar_like", "in30_like"):
        labels = g[f"subset/{name}/labels"]
        for mode in ("ovr", "loo", "ff"):
            normal = g[f"subset/{name}/{mode}/normal_classes"].tolist()
            assert data.normal_subset(labels, normal).tolist() == g[f"subset/{name}/{mode}/indices"].tolist()
            assert data.ad_targets(labels, normal).tolist() == batching.ad_targets(labels, normal).tolist()


def test_w8_kernel_keeps_out_of_the_accumulators(tmp_path):
    """gemm_w8.hip names its 128 accumulator registers literally (a[0:127] inside inline asm).  That is only sound while the compiler keeps
    out of the accumulator file: rebuild the file to assembly with the product's flags and require, per kernel, no AGPR reference and no
    v_accvgpr_* outside the asm statements, no scratch (a spilled fragment register would be stored before its LDS read has landed), and
    a kernel descriptor that allocates the 128 AGPRs."""
    import os
    import re
    import subprocess
    from eoe_amd import _build
    hipcc = _build.hipcc_path()
    if hipcc is None:
        pytest.skip("hipcc not available")
    src = os.path.join(_build.CSRC, "gemm_w8.hip")
    out = str(tmp_path / "w8.s")
    r = subprocess.run([hipcc] + _build._flags("gemm_w8.hip") + ["--cuda-device-only", "-S", "-x", "hip", src, "-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*gemm_w8_kernel\S*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 8, [k for k, _ in kernels]            # {fp16, bf16} x {plain, GELU} x {data-parallel, stream-K}
    for name, body in kernels:
        inside, bad = False, []
        for line in body.split("\n"):
            if ";;#ASMSTART" in line:
                inside = True
            elif ";;#ASMEND" in line:
                inside = False
            elif not inside:
                code = line.split(";")[0]
                if re.search(r"\ba\[?\d+", code) or "accvgpr" in code or "scratch_" in code:
                    bad.append(line.strip())
        assert not bad, (name, bad[:5])
        assert body.count("v_mfma_f32_16x16x32") == 6 * 32, name          # three iteration bodies of two 32-MFMA clusters
    for name, _ in kernels:
        assert re.search(r"\.set %s\.num_agpr, 128\b" % re.escape(name), text), name
        assert re.search(r"\.set %s\.private_seg_size, 0\b" % re.escape(name), text), name


def test_round5_entry_points_validate_their_arguments_without_a_gpu():
    """eoe_red_table_flush / eoe_probe_*: argument errors come back as error codes (no GPU is touched before the checks), an empty table is a no-op"""
    from eoe_amd import _lib
    assert _lib.lib.eoe_red_table_flush(None, None) == 1 and b"bad table" in _lib.lib.eoe_last_error()
    t = _lib.RedTable()
    t.count = 0
    assert _lib.lib.eoe_red_table_flush(C.byref(t), None) == 0
    t.count = _lib.RED_TABLE_MAX + 1
    assert _lib.lib.eoe_red_table_flush(C.byref(t), None) == 1
    assert _lib.lib.eoe_probe_mfma_f16(None, 0, 16, None) == 1
    assert _lib.lib.eoe_probe_copy(None, None, 1024, None) == 1
    assert C.sizeof(_lib.RedJob) == 48 and C.sizeof(_lib.RedTable) == 48 * _lib.RED_TABLE_MAX + 8
    # the gemm argument block's new tail (ABI v5): the stream-K / split-k workspace
    g = _lib.GemmArgs()
    assert hasattr(g, "sk_workspace") and hasattr(g, "sk_workspace_bytes") and _lib.ABI_VERSION == 5
