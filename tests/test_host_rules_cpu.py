"""Host-side rules that need no GPU: the kernel-choice rule of ops.linear (pure speed rule) and the reduction of
the two rocprofv3 PMC passes to profiles/pmc_traffic.json (tools/pmc_traffic.py): every launch of the dominant GEMM
instantiation under its plain key, its GEMM-API and pointwise-convolution populations apart, FETCH_SIZE doubled."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k256_choice_rule():
    from gomatching_amd import ops
    long_before = ops.K256_LONG
    try:
        ops.K256_LONG = True
        # short problems (the decoder's Q side): N = 256, or anything with a second addend
        assert ops.k256_wins(20000, 256, False) and ops.k256_wins(20000, 512, True) and not ops.k256_wins(20000, 512, False)
        assert ops.k256_wins(ops.K256_MAX_ROWS, 256, False)
        # long problems: wide outputs on the whole-line-store form, N = 256 stays on the tile kernel
        assert ops.k256_wins(297368, 640, False) and ops.k256_wins(297368, 1536, False) and not ops.k256_wins(297368, 256, False)
        assert not ops.k256_wins(297368, 256, True)
        ops.K256_LONG = False                                # the A/B switch restores the mid-round rule
        assert not ops.k256_wins(297368, 640, False) and ops.k256_wins(297368, 1536, False)
    finally:
        ops.K256_LONG = long_before


def _pmc_module():
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_reduction_splits_the_dominant_kernels_populations(tmp_path):
    mod = _pmc_module()
    dom = "void (anonymous namespace)::gemm_f16x3_kernel<128, 128, 0, 0, 3>((anonymous namespace)::Args)"
    rows = [(dom, 1024, 100.0), (dom, 1024, 300.0), (dom, 4096, 1000.0),            # two GEMM-API launches, one pointwise conv
            ("(anonymous namespace)::ffn_fused_kernel((anonymous namespace)::FfnArgs)", 512, 50.0)]
    for name, counter in (("f.csv", "FETCH_SIZE"), ("w.csv", "WRITE_SIZE")):
        with open(tmp_path / name, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            for kn, grid, val in rows:
                wr.writerow([kn, grid, counter, val])
                wr.writerow([kn, grid, "SOMETHING_ELSE", 7.0])
    fetch = mod.per_kernel(str(tmp_path / "f.csv"), "FETCH_SIZE", {1024})
    key = "gemm_f16x3_kernel<128,128,0,0,3>"
    assert fetch[key] == [100.0, 300.0, 1000.0]
    assert fetch[key + " [gemm api]"] == [100.0, 300.0] and fetch[key + " [pointwise conv]"] == [1000.0]
    assert fetch["ffn_fused_kernel"] == [50.0]
    assert mod.per_kernel(str(tmp_path / "f.csv"), "FETCH_SIZE", set())[key] == [100.0, 300.0, 1000.0]   # no grid file: no split
    # the hash that ties the committed counters to a build is the one bench.py checks
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert mod.kernel_source_hash() == bench.kernel_source_hash()
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        rec = json.load(f)
    assert "kernel_source_hash" in rec["_meta"] and key in rec and (key + " [gemm api]") in rec
    if rec["_meta"]["kernel_source_hash"] != bench.kernel_source_hash():
        assert bench.pmc_traffic(key) is None                # counters of another build are never printed
    else:
        assert bench.pmc_traffic(key) == rec[key]["hbm_bytes_per_launch"]


def test_builtin_configs_cover_the_reference_yamls():
    """The eight builtin configs = the inference-path keys of the reference's eight configs/*.yaml (values transcribed from them; when
    /root/reference is present -- the build container -- the yamls themselves are parsed and compared)."""
    import os
    from gomatching_amd.config import BUILTIN, REFERENCE_YAML, setup_cfg
    want = {  # head, rescoring, queries, vocabulary, score threshold, min / max size, NMS threshold
        "icdar15": ("LSTMatcher", True, 100, 37, 0.3, 1000, 3000, 0.5), "pp_icdar15": ("SHA_FFN_CRSATTN", True, 100, 37, 0.3, 1000, 3000, 0.5),
        "dstext": ("LSTMatcher", False, 300, 37, 0.5, 1280, 3000, 0.3), "pp_dstext": ("SHA_FFN_CRSATTN", False, 300, 37, 0.5, 1280, 3000, 0.3),
        "bovtext": ("LSTMatcher", False, 100, 5462, 0.5, 1000, 2400, 0.3), "pp_bovtext": ("SHA_FFN_CRSATTN", False, 100, 5462, 0.5, 1000, 2400, 0.3),
        "artvideo": ("LSTMatcher", True, 100, 37, 0.5, 1280, 3000, 0.3), "pp_artvideo": ("SHA_FFN_CRSATTN", True, 100, 37, 0.5, 1280, 3000, 0.3)}
    assert sorted(BUILTIN) == sorted(want) == sorted(REFERENCE_YAML)

    def keys(c):
        T = c.MODEL.TRANSFORMER
        return (c.MODEL.ROI_HEADS.NAME, bool(c.MODEL.ROI_HEADS.WITH_RESR), T.NUM_QUERIES, T.VOC_SIZE, T.INFERENCE_TH_TEST,
                c.INPUT.MIN_SIZE_TEST, c.INPUT.MAX_SIZE_TEST, c.VIDEO_TEST.NMS_THRESH)

    def more(c):
        A = c.MODEL.ASSO_HEAD
        return (A.ASSO_THRESH_TEST, A.NUM_WEIGHT_LAYERS, A.NUM_FC, A.NO_POS_EMB, c.INPUT.VIDEO.TEST_LEN, c.INPUT.FORMAT,
                tuple(sorted(dict(c.VIDEO_TEST).items())))
    for name, w in want.items():
        c = setup_cfg(builtin=name)
        assert keys(c) == w, name
        ref = os.path.join("/root/reference/configs", REFERENCE_YAML[name])
        if os.path.exists(ref):
            r = setup_cfg(ref)
            assert keys(r) == w and more(r) == more(c), name


def test_tail_form_rule_takes_occupancy_for_short_launches_and_throughput_for_long_ones():
    """ops.tail_form2_wins: the CU-cooperative tail (80-row workgroups, ~108 us each) against the round-5 form (128 rows, ~148 us):
    one round of the chip at the BASELINE decoder's 8 x 100 x 25 rows -> form 2; 8 x 300 x 25 rows (configs[3]) = three rounds of
    form 2 against two of form 1 -> form 1."""
    from gomatching_amd import ops
    assert ops.tail_form2_wins(8 * 100 * 25) and ops.tail_form2_wins(2500) and ops.tail_form2_wins(1)
    assert not ops.tail_form2_wins(8 * 300 * 25)
    assert ops.tail_form2_wins(16 * 100 * 25)
