"""CPU-side checks of the host logic: registry / module tree / state_dict parity with the oracle, the C-ABI library
loads and exports every declared symbol, layouts and path tables, and the product refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    from equiformer_amd import lib
    header = open(os.path.join(ROOT, "include", "equiformer_hip.h")).read()
    declared = set(re.findall(r"\b(eqf_[a-z0-9_]+)\s*\(", header))
    declared -= {"eqf_irreps", "eqf_rows", "eqf_dtp_paths"}
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(hip_lib, name), name
    assert declared - {"eqf_version"} == set(lib.SIGNATURES), "ctypes table out of sync with the header"
    assert lib.version().startswith("equiformer_hip")
    # and the other way round: every eqf_* symbol the shared object exports is declared -- in the public header or, for the
    # development switches, in equiformer_hip_dev.h (which the product never binds)
    import subprocess
    dev = set(re.findall(r"\b(eqf_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "equiformer_hip_dev.h")).read()))
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("eqf_")}
    # helper symbols shared between the translation units of the library (not entry points)
    internal = {n for n in exported if n.startswith("eqf_prof_begin") or n.startswith("eqf_prof_end")}
    assert exported - internal <= declared | dev, sorted(exported - internal - declared - dev)
    assert not (dev & set(lib.SIGNATURES)), "development switches must stay out of the product's binding table"


def test_binding_table_matches_header_prototypes():
    """Every ctypes signature has as many arguments as the C prototype it binds (the table is written by hand)."""
    import re
    from equiformer_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "equiformer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(?:int|long)\s+(eqf_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)}
    assert len(protos) > 60
    checked = 0
    for name, argtypes in lib.SIGNATURES.items():
        assert name in protos, name
        args = protos[name].strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        assert n == len(argtypes), (name, n, len(argtypes))
        checked += 1
    assert checked == len(lib.SIGNATURES) and checked > 60


def test_struct_layouts_match_header():
    from equiformer_amd import lib
    assert ctypes.sizeof(lib.EqfIrreps) == 4 * (1 + 3 * lib.EQF_MAX_SEG)  # nseg, l[], mul[], odd[]
    assert ctypes.sizeof(lib.EqfDtpPaths) == 4 * (6 + 11 * lib.EQF_MAX_PATHS)
    assert ctypes.sizeof(lib.EqfRows) == 12


def test_argument_errors_are_reported_not_crashed(hip_lib):
    from equiformer_amd import lib
    rows = lib.EqfRows(1, 4, 0)
    assert hip_lib.eqf_gemm_nn(None, rows, None, 4, None, rows, None, 4, 4, 4, 0, None) == -1
    assert hip_lib.eqf_lnsilu_fwd(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 4, 128,
                                  1e-5, None) == -2
    with pytest.raises(lib.HipLibraryError):
        lib.call("eqf_layernorm_fwd", None, None, None, None, None, None, 4, None, 1e-5, None)


def test_registry_and_factories():
    from equiformer_amd import nets
    for name in ["graph_attention_transformer_nonlinear_l2", "graph_attention_transformer_nonlinear_exp_l2_md17",
                 "graph_attention_transformer_nonlinear_exp_l3_md17", "graph_attention_transformer_l2",
                 "graph_attention_transformer_nonlinear_l2_e3", "graph_attention_transformer_nonlinear_bessel_l2"]:
        assert callable(nets.model_entrypoint(name))
    with pytest.raises(KeyError):
        nets.model_entrypoint("no_such_model")
    # variants outside the hot path fail loudly instead of silently degrading
    m = nets.model_entrypoint("graph_attention_transformer_l2")("5x0e", 5.0)  # linear-message variant is built
    assert sum(p.numel() for p in m.parameters()) == 3008515 and hasattr(m.blocks[0].ga, "sep")
    mb = nets.model_entrypoint("graph_attention_transformer_nonlinear_bessel_l2")("5x0e", 5.0)  # Bessel basis: built
    assert "rbf.rbf.frequencies" in dict(mb.named_parameters()) and "rbf.rbf.frequencies" in mb.no_weight_decay()
    # 128 frequencies replace the Gaussian layer's mean / std / weight / bias (2 * 128 + 2)
    assert sum(p.numel() for p in mb.parameters()) == 3531715 - (2 * 128 + 2) + 128
    for name in ("graph_attention_transformer_l2_md17", "graph_attention_transformer_nonlinear_bessel_l2_md17",
                 "graph_attention_transformer_nonlinear_bessel_l3_md17", "graph_attention_transformer_nonlinear_l2_md17"):
        assert callable(nets.model_entrypoint(name))
    # E(3) (parity-aware) irreps: built, on the un-fused tensor-product kernels (the fused ones key on the degree only)
    me = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2_e3")("5x0e", 5.0)
    assert sum(p.numel() for p in me.parameters()) == 3282211
    t = me.blocks[0].ga.sep_act.dtp.table
    assert t.has_odd and not t.fusable and len(t.paths) == 30 and not me.blocks[0].ga.act_sfc_spec.supported
    assert repr(t.irreps_out) == "176x0e+80x0o+160x1e+256x1o+240x2e+144x2o"
    assert me.blocks[0].norm_1.affine_bias.numel() == 128  # bias on 0e only
    for name in ("graph_attention_transformer_nonlinear_l2_e3_md17", "graph_attention_transformer_nonlinear_exp_l3_e3_md17",
                 "graph_attention_transformer_nonlinear_bessel_l3_e3_md17", "oc20_l1_256_e3_nonlinear"):
        assert callable(nets.model_entrypoint(name))


def test_oc20_and_md17_head_variants_build_with_reference_keys():
    """Auxiliary / attention heads and stochastic depth (OC20 *_aux_* configs, MD17 ..._attn_exp_l3_md17): same parameter
    names and shapes as the oracle's restatement of the reference, heads included in the radial bank."""
    from equiformer_amd import nets
    from equiformer_amd.nets.layers import GraphDropPath
    from oracle import nets as onets
    over = dict(num_layers=2)
    m = nets.model_entrypoint("oc20_l1_256_nonlinear_aux")(**over)
    o = onets.oc20_l1_256_nonlinear(irreps_feature="512x0e+256x1e", use_auxiliary_task=True, drop_path_rate=0.05, **over)
    assert {k: v.shape for k, v in m.state_dict().items()} == {k: v.shape for k, v in o.state_dict().items()}
    assert isinstance(m.blocks[0].drop_path, GraphDropPath) and m.blocks[0].drop_path.drop_prob == 0.05
    assert m.head[0].layout_out.dim == 512 and "auxiliary_head.alpha_dot" in m.state_dict()
    assert len(m._radial_bank().modules) == 1 + 2 + 1  # degree embedding, blocks, auxiliary head
    m = nets.model_entrypoint("oc20_graph_attention_transformer")(
        name="graph_attention_transformer", num_layers=1, use_attention_head=True, use_auxiliary_task=True)
    assert m.head_skip_connect.layout_out.dim == 4 and not hasattr(m, "auxiliary_head")
    assert nets.model_entrypoint("oc20_l1_256")(num_layers=1).blocks[0].ga.nonlinear_message is False
    assert len(nets.model_entrypoint("oc20_l1_256_blocks18_nonlinear_aux")().blocks) == 18
    m = nets.model_entrypoint("graph_attention_transformer_nonlinear_attn_exp_l3_md17")("64x0e", 5.0, num_basis=32)
    o = onets.GraphAttentionTransformerMD17(
        irreps_in="64x0e", irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", irreps_sh="1x0e+1x1e+1x2e+1x3e",
        max_radius=5.0, number_of_basis=32, basis_type="exp", irreps_feature="128x0e+64x1e+64x2e+32x3e",
        irreps_head="32x0e+16x1e+16x2e+8x3e", num_heads=4, nonlinear_message=True,
        irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e", use_attn_head=True, alpha_drop=0.0)
    assert {k: v.shape for k, v in m.state_dict().items()} == {k: v.shape for k, v in o.state_dict().items()}
    with pytest.raises(NotImplementedError):
        nets.model_entrypoint("oc20_l1_256_nonlinear")(use_atom_edge_attr=True)
    # DeNS: module (= parameter) order of the reference, names shared with the oracle
    m = nets.model_entrypoint("equiformer_md17_dens_l2")(num_layers=1)
    assert [n for n, _ in m.named_children()] == ["atom_embed", "rbf", "edge_deg_embed", "force_embed", "blocks", "norm",
                                                  "energy_head", "scale_scatter", "denoising_pos_head"]
    o = onets.Equiformer_MD17_DeNS(
        num_layers=1, irreps_mlp_mid="384x0e+192x1e+96x2e", irreps_head="32x0e+16x1e+8x2e")
    assert {k: v.shape for k, v in m.state_dict().items()} == {k: v.shape for k, v in o.state_dict().items()}
    assert len(nets.model_entrypoint("equiformer_md17_dens_l3")(num_layers=1).denoising_pos_head.heads_layout.segs) == 4
    for name in ("dot_product_attention_transformer_exp_l3_md17", "oc20_dp_l1_256", "oc20_dp_attention_transformer",
                 "equiformer_md17_dens"):
        assert callable(nets.model_entrypoint(name))


@pytest.mark.parametrize("name,kw,count", [
    ("graph_attention_transformer_nonlinear_l2", dict(irreps_in="5x0e", radius=5.0), 3531715),
    ("graph_attention_transformer_nonlinear_exp_l2_md17", dict(irreps_in="64x0e", radius=5.0, num_basis=32), 3496001),
    ("graph_attention_transformer_nonlinear_exp_l3_md17", dict(irreps_in="64x0e", radius=5.0, num_basis=32), 5500865),
    ("oc20_l1_256_nonlinear", dict(), 9123331),
    ("dot_product_attention_transformer_l2", dict(irreps_in="5x0e", radius=5.0), 3352579),
    ("dot_product_attention_transformer_exp_l2_md17", dict(irreps_in="64x0e", radius=5.0, num_basis=32), 3316865)])
def test_parameter_counts_and_state_dict_keys(name, kw, count):
    from equiformer_amd import nets
    from oracle import nets as onets
    m = nets.model_entrypoint(name)(**kw)
    assert sum(p.numel() for p in m.parameters()) == count
    ofac = dict(onets.ENTRYPOINTS, oc20_l1_256_nonlinear=lambda **k: onets.oc20_l1_256_nonlinear())[name]
    o = ofac(**kw) if kw else ofac()
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm.keys()) == list(so.keys())
    for k in sm:
        assert sm[k].shape == so[k].shape, k
    if hasattr(m, "no_weight_decay"):
        nwd = m.no_weight_decay()
        assert "blocks.0.norm_1.affine_weight" in nwd and ("rbf.mean" in nwd or "md17" in name or "oc20" in name)
        rad = "blocks.0.dpa.key_value.dtp_rad" if "dot_product" in name else "blocks.0.ga.sep_act.dtp_rad"
        assert rad + ".net.0.bias" in nwd and rad + ".net.0.weight" not in nwd


def test_appendix_a_keys_present():
    from equiformer_amd import nets
    sd = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")("5x0e", 5.0).state_dict()
    expect = {"atom_embed.atom_type_lin.tp.weight": (640,), "atom_embed.atom_type_lin.bias.0": (128,),
              "rbf.mean": (1, 128), "rbf.weight": (1, 1), "edge_deg_embed.exp.tp.weight": (128,),
              "edge_deg_embed.rad.net.6.weight": (960, 64), "edge_deg_embed.rad.offset": (960,),
              "edge_deg_embed.proj.tp.weight": (64512,), "blocks.0.norm_1.affine_weight": (224,),
              "blocks.0.norm_1.affine_bias": (128,), "blocks.0.ga.merge_src.tp.weight": (21504,),
              "blocks.0.ga.merge_src.bias.0": (128,), "blocks.0.ga.merge_dst.tp.weight": (21504,),
              "blocks.0.ga.sep_act.lin.tp.weight": (86016,), "blocks.0.ga.sep_act.lin.bias.0": (224,),
              "blocks.0.ga.sep_alpha.tp.weight": (28672,), "blocks.0.ga.sep_value.dtp.tp.weight": (960,),
              "blocks.0.ga.sep_value.lin.tp.weight": (64512,), "blocks.0.ga.alpha_dot": (1, 4, 32),
              "blocks.0.ga.proj.tp.weight": (21504,), "blocks.0.ffn.fctp_1.tp.weight": (101376,),
              "blocks.0.ffn.fctp_1.bias.0": (672,), "blocks.4.ffn.fctp_2.tp.weight": (64512,),
              "blocks.5.ffn.fctp_2.tp.weight": (196608,), "blocks.5.ffn_shortcut.tp.weight": (65536,),
              "norm.affine_weight": (512,), "head.0.tp.weight": (262144,), "head.2.tp.weight": (512,),
              "head.2.bias.0": (1,)}
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert "blocks.0.ga.merge_dst.bias.0" not in sd


def test_no_cpu_fallback():
    from equiformer_amd import nets, ops
    m = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")("5x0e", 5.0)
    pos = torch.rand(6, 3)
    with pytest.raises(ops.HipOnlyError):
        m(None, pos, torch.zeros(6, dtype=torch.long), torch.tensor([1, 6, 6, 8, 1, 1]))
    with pytest.raises(ops.HipOnlyError):
        ops.dense_linear(torch.rand(4, 4), torch.rand(4, 4), None)


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; import equiformer_amd, equiformer_amd.nets, equiformer_amd.ops, equiformer_amd.graph, "
            "equiformer_amd.parallel, equiformer_amd.synthetic; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "equiformer_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_layout_permutations_roundtrip():
    from equiformer_amd.layout import RowLayout
    lay = RowLayout("8x0e+4x1e+2x2e")
    assert lay.dim == 8 + 12 + 10 and lay.offsets == [0, 8, 20]
    x = torch.arange(lay.dim, dtype=torch.float32)[None]
    cf = x[:, lay.perm_from_e3nn()]
    assert torch.equal(cf[:, lay.perm_to_e3nn()], x)
    # e3nn element (seg 1, u=2, m=1) sits at 8 + 2*3 + 1; in CF at 8 + 1*4 + 2
    assert cf[0, 8 + 1 * 4 + 2].item() == 8 + 2 * 3 + 1
    e3 = RowLayout("8x0e+2x0o+4x1e+4x1o")  # E(3) rows: one segment per (degree, parity), even first
    assert e3.par == [1, -1, 1, -1] and e3.seg_index(1, -1) == 3 and e3.mul_of(0, -1) == 2 and e3.c.odd[1] == 1
    with pytest.raises(NotImplementedError):
        RowLayout("8x0e+4x1o+4x1e")  # odd before even: not the sorted order every reference model uses


def test_dtp_table_matches_oracle_instructions():
    from equiformer_amd.layout import DtpTable
    from oracle import nets as onets
    for irr, sh in [("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"), ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e"),
                    ("256x0e+128x1e", "1x0e+1x1e")]:
        t = DtpTable(irr, sh, irr)
        o = onets.DepthwiseTensorProduct(irr, sh, irr, bias=False)
        assert t.weight_numel == o.tp.weight_numel and len(t.paths) == len(o.tp.instructions)
        assert repr(t.irreps_out) == repr(o.irreps_out.simplify())
        assert repr(t.irreps_out_unsimplified) == repr(o.irreps_out)
        slices = o.irreps_out.slices()
        for p, (i1, i2, io, *_rest) in zip(t.paths, o.tp.instructions):
            assert (p["l1"], p["l2"], p["l3"]) == (o.irreps_in1[i1][1].l, o.irreps_in2[i2][1].l, o.irreps_out[io][1].l)
            # first element of the path's output slice in e3nn order <-> (out_off, out_ch) in CF order
            seg_start = sum(m * ir.dim for m, ir in o.irreps_out.simplify() if ir.l < p["l3"])
            assert p["out_off"] == seg_start
            assert slices[io].start == seg_start + p["out_ch"] * (2 * p["l3"] + 1)
    assert DtpTable("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e").m_numel == 179


def test_dtp_table_with_parity_matches_oracle_instructions():
    """E(3) irreps: paths keyed on (degree, parity); spherical harmonics carry (-1)^l; only 0e is forced."""
    from equiformer_amd.layout import DtpTable
    from oracle import nets as onets
    cases = [("128x0e+32x0o+32x1e+32x1o+16x2e+16x2o", "1x0e+1x1o+1x2e", 30),
             ("128x0e+64x0o+32x1e+32x1o+32x2e+32x2o+16x3e+16x3o", "1x0e+1x1o+1x2e+1x3o", 68),
             ("256x0e+64x0o+64x1e+64x1o", "1x0e+1x1o", 10)]
    for irr, sh, npaths in cases:
        t = DtpTable(irr, sh, irr)
        o = onets.DepthwiseTensorProduct(irr, sh, irr, bias=False)
        assert len(t.paths) == len(o.tp.instructions) == npaths and t.weight_numel == o.tp.weight_numel
        assert repr(t.irreps_out) == repr(o.irreps_out.simplify())
        assert repr(t.irreps_out_unsimplified) == repr(o.irreps_out)
        assert t.has_odd and not t.fusable
        slices = o.irreps_out.slices()
        simp = list(o.irreps_out.simplify())
        for p, (i1, i2, io, *_rest) in zip(t.paths, o.tp.instructions):
            ir1, ir2, iro = o.irreps_in1[i1][1], o.irreps_in2[i2][1], o.irreps_out[io][1]
            assert (p["l1"], p["l2"], p["l3"], p["p3"]) == (ir1.l, ir2.l, iro.l, iro.p) and iro.p == ir1.p * ir2.p
            seg_start = 0
            for m, ir in simp:
                if (ir.l, ir.p) == (iro.l, iro.p):
                    break
                seg_start += m * ir.dim
            assert p["out_off"] == seg_start
            assert slices[io].start == seg_start + p["out_ch"] * (2 * p["l3"] + 1)
    # an SE(3)-flavoured table (all-even harmonics) couples 1e x 1e -> 1e; the E(3) one sends 1o x 1o to 1e, never to 1o
    e3 = DtpTable("8x1o", "1x0e+1x1o", "8x0e+8x1e+8x1o")
    assert sorted((p["l2"], p["l3"], p["p3"]) for p in e3.paths) == [(0, 1, -1), (1, 0, 1), (1, 1, 1)]


def test_so3_matches_oracle_and_constants():
    import numpy as np
    from equiformer_amd import so3
    from oracle import e3, nets as onets
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
                assert np.abs(so3.wigner_3j(l1, l2, l3) - e3.wigner_3j(l1, l2, l3).numpy()).max() < 1e-12
    assert abs(so3.C_SILU - e3.normalize2mom_const(torch.nn.functional.silu)) < 1e-12
    assert abs(so3.C_SIGMOID - e3.normalize2mom_const(torch.sigmoid)) < 1e-12
    assert abs(so3.C_SMOOTH_LEAKY_RELU_02 - e3.normalize2mom_const(onets.SmoothLeakyReLU(0.2))) < 1e-12


def test_synthetic_batches_have_the_survey_statistics():
    from equiformer_amd.synthetic import md17_aspirin_batch, qm9_like_batch
    from oracle import nets as onets
    d = qm9_like_batch(32, 18, side=6.5, seed=0)
    src, _ = onets.radius_graph(d["pos"], 5.0, d["batch"])
    assert 170 < src.numel() / 32 < 230          # "~200 edges" per molecule (198 +- 21)
    d = qm9_like_batch(32, 18, side=5.0, seed=0)
    src, _ = onets.radius_graph(d["pos"], 5.0, d["batch"])
    assert 255 < src.numel() / 32 < 300          # QM9 statistics point (~277)
    a = md17_aspirin_batch(2)
    assert a["pos"].shape == (42, 3) and sorted(a["z"][:21].tolist()) == [1] * 8 + [6] * 9 + [8] * 4
    src, _ = onets.radius_graph(a["pos"], 5.0, a["batch"])
    assert 320 < src.numel() / 2 <= 420
    dmin = torch.cdist(a["pos"][:21], a["pos"][:21]) + 10 * torch.eye(21)
    assert dmin.min() > 0.7


def test_model_can_be_deep_copied_and_pickled():
    """The drivers deep-copy the model for the EMA (timm ModelEma, main_qm9.py:169-175) and torch.save whole models."""
    import copy
    import io
    from equiformer_amd import nets
    m = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0, num_basis=32)
    m2 = copy.deepcopy(m)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in m2.named_parameters()]
    buf = io.BytesIO()
    torch.save(m, buf)
    assert buf.tell() > 0


def test_split_precision_kernels_are_not_chosen_for_launches_beyond_32_bit_offsets():
    """eqf_sfcx_supported plans with a nominal E, the kernels index per-edge tensors with 32-bit element offsets: a
    launch whose widest tensor reaches 2^31 elements must be routed to the exact-fp32 kernels, not fail in the planner."""
    from equiformer_amd import ops
    from equiformer_amd.layout import DtpTable, RowLayout
    table = DtpTable("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e")
    spec = ops.SfcSpec(table, RowLayout("224x0e+64x1e+32x2e"), n2=128)
    assert spec.x_mask(0) == 7 and spec.x_mask(0, 25354) == 7
    big = (1 << 31) // spec._widest_row() + 1
    assert spec.x_mask(0, big) == 0 and spec.x_mask(0, big - 2) == 7
    prev = ops.set_matrix_mode("split")
    try:
        assert ops._sfc_mode(spec, 25354) == 0 and ops._sfc_mode(spec, big) is None
        assert not ops.sep_fctp_gated_ok(ops.SfcSpec(table, RowLayout("128x0e+64x1e+32x2e")), 576, 128,
                                         RowLayout("64x1e+32x2e"), E=big)
    finally:
        ops.set_matrix_mode(prev)
