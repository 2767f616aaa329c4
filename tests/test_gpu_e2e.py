"""GPU parity of the whole path through the drop-in modules (pmce_amd.models) vs (a) the reference's own outputs
committed as golden fixtures and (b) the oracle on fresh seeded inputs; plus size-independent properties at
BASELINE.json's full batch sizes.  Contract (north_star): <= 1e-3 max-abs on fp32 vertices/joints in metres,
bit-exact integer gather."""
import numpy as np
import pytest
import torch

from conftest import cached_state_dict

pytestmark = pytest.mark.gpu

TOL_M = 1e-3          # metres: the contract
TIGHT_M = 5e-5        # what fp32 re-association actually gives (oracle fp32-vs-fp64 noise floor is ~2e-6 m)
TOL_MM = 2e-2         # pose3d is in millimetres with |values| ~1e3 (fp32 ulp there = 6e-5 mm)


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def maxabs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


_MODELS = {}


def get_model(J, C):
    from pmce_amd import assets, models
    key = (J, C)
    if key not in _MODELS:
        _MODELS.clear()
        torch.cuda.empty_cache()
        m = models.PMCE.get_model(J, C, 3)
        m.load_state_dict(cached_state_dict(J, C))
        m.set_j_regressor(assets.load_j_regressor("h36m"))
        _MODELS[key] = m.to(dev())
    return _MODELS[key]


@pytest.mark.parametrize("mode", ["default", "split_f16 at every batch size", "f32"])
@pytest.mark.parametrize("name", ["e2e_J17_C256_B2.npz", "e2e_J19_C256_B1.npz", "e2e_J17_C512_B1.npz"])
def test_forward_matches_reference_fixture(golden, name, mode):
    """Outputs of the reference's own forward (fixtures made by importing it, tests/golden/make_golden_e2e.py) - in the default
    configuration (small batches stay on the fp32 pipe), with the three-product f16 form forced, and on the fp32 pipe."""
    from pmce_amd import synth
    z = golden(name)
    J, C, B = int(z["J"]), int(z["C"]), int(z["B"])
    model = get_model(J, C)
    if mode == "default":
        model.set_gemm_mode(None)
    elif mode == "f32":
        model.set_gemm_mode("f32")
    else:
        model.set_gemm_mode("split_f16", min_batch=1)
    assert np.array_equal(model.vj_relation, z["vj_relation"])
    pose2d, img_feat = synth.make_inputs(B, J, int(z["input_seed"]))
    try:
        mesh, pose, pose3d, pred = model.forward_with_joints(T(pose2d).to(dev()), T(img_feat).to(dev()))
        torch.cuda.synchronize()
        eng = model._engine
        inter = {key: buf.clone() for key, buf in (("g_mid", eng.intermediate("g", B, (B, 2048))), ("v1", eng.intermediate("VT1", B, (B, 431, 3))),
                                                     ("v2", eng.intermediate("VT2", B, (B, 431, 3))), ("v3", eng.intermediate("VT0", B, (B, 431, 3))))}
    finally:
        model.set_gemm_mode(None)          # the cached model goes back to the library's defaults
    e = dict(mesh=maxabs(mesh, T(z["cam_mesh"])), pose=maxabs(pose, T(z["cam_pose"])),
             pose3d_mm=maxabs(pose3d, T(z["pose3d"])), pred_mm=maxabs(pred, T(z["pred_pose"])))
    print(name, {k: f"{v:.2e}" for k, v in e.items()})
    for key, buf in inter.items():
        ei = maxabs(buf, T(z[key]))
        print(f"   intermediate {key}: {ei:.2e}")
        assert ei < TIGHT_M, key
    assert e["mesh"] < TOL_M and e["pose"] < TOL_M              # the contract
    assert e["mesh"] < TIGHT_M and e["pose"] < TIGHT_M          # and what we actually hold ourselves to
    assert e["pose3d_mm"] < TOL_MM and e["pose3d_mm"] / 1000 < TOL_M
    assert e["pred_mm"] < 1000 * TIGHT_M


def test_forward_matches_oracle_fresh_inputs():
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    J, C, B = 17, 256, 3
    model = get_model(J, C)
    sd = cached_state_dict(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 77)
    mesh, pose, pose3d = model(T(pose2d).to(dev()), T(img_feat).to(dev()))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation)
    e = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    print("fresh inputs vs oracle: mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM


def test_submodule_entry_points():
    """models.PoseEstimation.get_model (LiftTester path) and models.CoevoDecoder.get_model stand alone."""
    from oracle import pmce_oracle as O
    from pmce_amd import models, synth
    J = 17
    sd = cached_state_dict(J, 256)
    lifter = models.PoseEstimation.get_model(J, 256, 3)
    lifter.load_state_dict({k[len("pose_lifter."):]: v for k, v in sd.items() if k.startswith("pose_lifter.")})
    lifter = lifter.to(dev())
    pose2d, img_feat = synth.make_inputs(2, J, 9)
    out = lifter(T(pose2d).to(dev()), T(img_feat).to(dev()))
    with torch.no_grad():
        ref = O.lifter_forward(sd, T(pose2d), T(img_feat))
    assert maxabs(out, ref) < TOL_MM
    dec = models.CoevoDecoder.get_model(J, 256)
    dec.load_state_dict({k[len("pose_mesh_coevo."):]: v for k, v in sd.items() if k.startswith("pose_mesh_coevo.")})
    dec = dec.to(dev())
    joints, feats = synth.make_decoder_inputs(2, J, 4)
    pose, mesh = dec(T(joints).to(dev()), T(feats).to(dev()))
    with torch.no_grad():
        rj, rm = O.decoder_forward(sd, T(joints), T(feats), dec.vj_relation, "pose_mesh_coevo.")
    e = (maxabs(pose, rj), maxabs(mesh, rm))
    print("decoder-only vs oracle: pose %.2e mesh %.2e" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M


def test_checkpoint_forms_and_errors(tmp_path):
    from pmce_amd import _lib, models
    sd = cached_state_dict(17, 256)
    m = models.PMCE.get_model(17, 256, 3)
    wrapped = {"epoch": 1, "model_state_dict": {"module." + k: v for k, v in sd.items()}}   # DataParallel-saved checkpoint
    m.load_state_dict(wrapped)
    assert set(m.state_dict().keys()) == set(sd.keys())
    with pytest.raises(_lib.PmceError):
        m(torch.zeros(1, 16, 17, 2), torch.zeros(1, 16, 2048))          # CPU model: no fallback
    m = m.to(dev())
    with pytest.raises(_lib.PmceError):
        m(torch.zeros(1, 16, 17, 2), torch.zeros(1, 16, 2048))          # CPU inputs: no fallback
    with pytest.raises(ValueError):
        m(torch.zeros(1, 16, 19, 2, device=dev()), torch.zeros(1, 16, 2048, device=dev()))
    with pytest.raises(NotImplementedError):
        m.train()


@pytest.mark.parametrize("B,C", [(64, 256), (256, 256), (256, 512)])
def test_full_size_properties(B, C):
    """BASELINE configs[1]/[2] sizes, at the reference's width and at north_star's C = 512: properties that need no oracle run.
    (1) clips are independent: clip i of a big batch == the same clip run in a batch of 2;
    (2) permuting clips permutes outputs; (3) J_regressor projection is linear in the mesh."""
    from pmce_amd import assets, ops, synth
    J = 17
    model = get_model(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 123)
    p, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    mesh, pose, pose3d, pred = model.forward_with_joints(p, f)
    assert torch.isfinite(mesh).all() and torch.isfinite(pose).all() and torch.isfinite(pose3d).all()
    idx = [0, B // 2, B - 1]
    for i in idx:
        m2, q2, l2 = model(p[[i, (i + 1) % B]], f[[i, (i + 1) % B]])
        assert maxabs(m2[0], mesh[i]) < 1e-5 and maxabs(q2[0], pose[i]) < 1e-5 and maxabs(l2[0], pose3d[i]) < 1e-2
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev())
    mp, qp, lp = model(p[perm], f[perm])
    assert maxabs(mp, mesh[perm]) < 1e-5 and maxabs(qp, pose[perm]) < 1e-5
    jr = assets.load_j_regressor("h36m")
    a = ops.j_regress(mesh, jr)
    assert maxabs(a, pred) == 0.0
    b = ops.j_regress(2.0 * mesh, jr)
    assert maxabs(b, 2.0 * a) < 1e-3
    # regressed joints against a dense fp64 product
    ref = torch.einsum("jv,bvl->bjl", T(jr).double(), mesh.double().cpu() * 1000)
    assert maxabs(pred, ref) < 5e-3


def test_streaming_windows_and_packed_weights(tmp_path):
    """BASELINE config 5 plumbing: GPU window assembly is a bit-exact gather, stream_forward == forward on explicit windows;
    and the packed-weights export round-trips."""
    from pmce_amd import checkpoint, streaming, synth
    J = 17
    model = get_model(J, 256)
    L = 40
    p_np, f_np = synth.make_inputs(3, J, 5)                      # 48 frames of per-frame data
    pose_fr = T(p_np.reshape(-1, J, 2)[:L]).to(dev())
    feat_fr = T(f_np.reshape(-1, 2048)[:L]).to(dev())
    win = streaming.demo_window_list(L)                          # includes repeated-frame head/tail windows
    wp, wf = streaming.assemble_windows(pose_fr, feat_fr, win)
    for w, (s, e) in enumerate(win):
        idx = [s] * 16 if s == e else list(range(s, e + 1))
        assert torch.equal(wf[w], feat_fr[idx]) and torch.equal(wp[w], pose_fr[idx])
    out = streaming.stream_forward(model, pose_fr, feat_fr, windows=win, batch=16)
    ref = model(wp, wf)
    assert out[0].shape == (L, 6890, 3)
    assert maxabs(out[0], ref[0]) < 1e-5 and maxabs(out[2], ref[2]) < 1e-2
    # packed weights file
    f = tmp_path / "packed.safetensors"
    meta = checkpoint.export_packed(model, str(f))
    tensors, meta2 = checkpoint.load_packed(str(f), dev())
    assert meta2["num_joint"] == "17" and set(tensors) == set(model._engine.packed)
    for k, v in tensors.items():
        assert torch.equal(v, model._engine.packed[k]), k


def test_empty_batch():
    """B = 0: empty outputs of the right shapes from all three modules (no launch), like the reference's modules."""
    from pmce_amd import models
    J = 19
    model = get_model(J, 256)
    mesh, pose, pose3d, pred = model.forward_with_joints(torch.zeros(0, 16, J, 2, device=dev()), torch.zeros(0, 16, 2048, device=dev()))
    assert mesh.shape == (0, 6890, 3) and pose.shape == (0, J, 3) and pose3d.shape == (0, J, 3) and pred.shape == (0, 17, 3)
    sd = cached_state_dict(J, 256)
    lifter = models.PoseEstimation.get_model(J, 256, 3)
    lifter.load_state_dict({k[len("pose_lifter."):]: v for k, v in sd.items() if k.startswith("pose_lifter.")})
    assert lifter.to(dev())(torch.zeros(0, 16, J, 2, device=dev()), torch.zeros(0, 16, 2048, device=dev())).shape == (0, J, 3)
    dec = models.CoevoDecoder.get_model(J, 256)
    dec.load_state_dict({k[len("pose_mesh_coevo."):]: v for k, v in sd.items() if k.startswith("pose_mesh_coevo.")})
    p0, m0 = dec.to(dev())(torch.zeros(0, J, 3, device=dev()), torch.zeros(0, 16, 2048, device=dev()))
    assert p0.shape == (0, J, 3) and m0.shape == (0, 6890, 3)


@pytest.mark.parametrize("B", [1, 77])
def test_odd_batch_sizes(B):
    """ragged batch sizes (no tile of any kernel is full): clip i of the batch == the same clip in a batch of 3."""
    from pmce_amd import synth
    J = 19
    model = get_model(J, 256)
    pose2d, img_feat = synth.make_inputs(B, J, 31)
    p, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    mesh, pose, pose3d = model(p, f)
    assert torch.isfinite(mesh).all()
    for i in sorted({0, B // 2, B - 1}):
        sel = [i, 0, B - 1]
        m2, q2, l2 = model(p[sel], f[sel])
        assert maxabs(m2[0], mesh[i]) < 1e-5 and maxabs(q2[0], pose[i]) < 1e-5 and maxabs(l2[0], pose3d[i]) < 1e-2


def test_streaming_frame_reuse_matches_window_forward():
    """serving windows from per-frame tables (first spatial block + GRU layer-0 projections computed once
    per frame) gives the same outputs as running every window as an independent clip."""
    import time
    from pmce_amd import streaming, synth
    J = 17
    model = get_model(J, 256)
    L = 16 * 20
    p_np, f_np = synth.make_inputs(20, J, 8)
    pose_fr = T(p_np.reshape(-1, J, 2)[:L]).to(dev())
    feat_fr = T(f_np.reshape(-1, 2048)[:L]).to(dev())
    win = streaming.demo_window_list(L)                       # L windows incl. repeated-frame head/tail
    ref = streaming.stream_forward(model, pose_fr, feat_fr, windows=win, batch=128, with_joints=True)
    cache = streaming.precompute_frames(model, pose_fr, feat_fr)
    out = streaming.stream_forward_cached(model, cache, windows=win, batch=128, with_joints=True)
    e = [maxabs(a, b) for a, b in zip(out, ref)]
    print("frame-reuse vs window forward: mesh %.2e pose %.2e pose3d %.2e mm pred %.2e mm" % tuple(e))
    assert e[0] < 1e-5 and e[1] < 1e-5 and e[2] < 1e-2 and e[3] < 1e-2
    # batches in flight on several lanes: identical to one lane
    one = streaming.stream_forward_cached(model, cache, windows=win, batch=64, with_joints=True, lanes=1)
    for lanes in (2, 3):
        many = streaming.stream_forward_cached(model, cache, windows=win, batch=64, with_joints=True, lanes=lanes)
        assert all(torch.equal(a, b) for a, b in zip(one, many))
    for fn, name in ((lambda: streaming.stream_forward(model, pose_fr, feat_fr, windows=win, batch=320), "independent windows"),
                     (lambda: streaming.stream_forward_cached(model, streaming.precompute_frames(model, pose_fr, feat_fr),
                                                              windows=win, batch=320), "frame reuse"),
                     (lambda: streaming.stream_forward_cached(model, streaming.precompute_frames(model, pose_fr, feat_fr),
                                                              windows=win, batch=160), "frame reuse, 2 x 160 in flight")):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f"   {name}: {5 * L / (time.perf_counter() - t0):.0f} windows/s")


def test_forward_is_graph_capturable():
    """The launch sequence (with its internal fork/join onto the side stream) has no host synchronisation or allocation:
    it can be captured into a HIP graph and replayed, bit-identically."""
    from pmce_amd import synth
    J, B = 17, 8
    model = get_model(J, 256)
    pose2d, img_feat = synth.make_inputs(B, J, 55)
    p, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            ref = model.forward_with_joints(p, f)          # warm-up outside capture: workspace, side stream, events
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = model.forward_with_joints(p, f)
    for o in out:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


def test_pipeline_matches_direct_forward():
    """Several batches in flight on shared weights (models.PMCE.Pipeline) == the same batches run one at a time, bit for bit,
    also when the batch size changes between submissions."""
    from pmce_amd import synth
    J = 17
    model = get_model(J, 256)
    batches = []
    for i, B in enumerate((5, 5, 3, 8, 5, 1)):
        p, f = synth.make_inputs(B, J, 700 + i)
        batches.append((T(p).to(dev()), T(f).to(dev())))
    direct = [tuple(t.clone() for t in model.forward_with_joints(p, f)) for p, f in batches]
    pipe = model.pipeline(depth=3)
    tickets = [pipe.submit(p, f) for p, f in batches]
    for d, t in zip(direct, tickets):
        got = t.result()
        for a, b in zip(d, got):
            assert torch.equal(a, b)
    pipe.synchronize()


@pytest.mark.parametrize("stagger", [None, True, False])
def test_pipeline_lane_forms_give_the_same_bits(stagger):
    """Pipeline(stagger=...): staggered lanes (a batch's lifter waits for the previous batch's), free-running lanes and the default (decided per
    submit from the batch size: staggered from 192 clips on, round 6) all return the direct forward's bits - also when batches of both kinds
    alternate through the same lanes."""
    from pmce_amd import synth
    from pmce_amd.models.PMCE import Pipeline
    J, C = 17, 256
    model = get_model(J, C)
    pipe = model.pipeline(depth=2, stagger=stagger)
    assert pipe.staggers(256) == (True if stagger is None else stagger) and pipe.staggers(8) == (False if stagger is None else stagger)
    assert Pipeline.STAGGER_FROM_BATCH == 192
    sizes = [5, 200, 3, 192, 191]
    batches = [synth.make_inputs(b, J, 40 + k) for k, b in enumerate(sizes)]
    dev_batches = [(T(p).to(dev()), T(f).to(dev())) for p, f in batches]
    tickets = [pipe.submit(p, f) for p, f in dev_batches]
    for (p, f), t in zip(dev_batches, tickets):
        ref = model.forward_with_joints(p, f)
        for a, b in zip(t.result(), ref):
            assert torch.equal(a, b)
    assert not pipe.synchronize()


def test_pipeline_lanes_share_packed_weights_and_split_batches():
    """Pipeline lanes in split_f16 mode: (1) batches large enough for the three-product f16 form come out bit-identical to direct
    forwards (the lanes overlap on their own streams); (2) a lane does not pack its own copy of the f16 weight planes (0.5 GB at
    C = 256): building three lanes costs workspaces only."""
    from pmce_amd import synth
    J, B = 17, 64
    model = get_model(J, 256)
    assert model.gemm_mode() == "split_f16"
    batches = []
    for i in range(3):
        p, f = synth.make_inputs(B, J, 900 + i)
        batches.append((T(p).to(dev()), T(f).to(dev())))
    direct = [tuple(t.clone() for t in model.forward_with_joints(p, f)) for p, f in batches]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    pipe = model.pipeline(depth=3).prepare(B)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    ws = model._engine.lib.pmce_model_workspace_bytes(model._engine.handle, B)
    print(f"three lanes at B={B}: {used / 2**20:.0f} MiB (one workspace is {ws / 2**20:.0f} MiB)")
    assert used < 3 * ws + (128 << 20)
    tickets = [pipe.submit(p, f) for p, f in batches]
    for d, t in zip(direct, tickets):
        for a, b in zip(d, t.result()):
            assert torch.equal(a, b)
    pipe.synchronize()


def test_overlapped_split_mode_equals_serial():
    """Kernels of the split-f16 mode overlap (two streams inside a forward, pipeline lanes on their own streams).  f16 matrix
    instructions disturb packed-fp32 arithmetic of co-resident waves on MI355X; the library contains none, and
    every kernel is deterministic - so overlapped runs must be BITWISE equal to the strictly serial run at full batch size."""
    from pmce_amd import _lib, synth
    assert _lib.split_overlap()
    J, B = 17, 256
    for C in (256, 512):
        model = get_model(J, C)
        model.set_gemm_mode("split_f16")
        batches = []
        for i in range(2):
            p, f = synth.make_inputs(B, J, 700 + i)
            batches.append((T(p).to(dev()), T(f).to(dev())))
        try:
            model.set_concurrency(False)
            serial = [tuple(t.clone() for t in model.forward_with_joints(p, f)) for p, f in batches]
            model.set_concurrency(True)
            for rep in range(12):
                got = model.forward_with_joints(*batches[rep % 2])
                assert all(torch.equal(a, b) for a, b in zip(got, serial[rep % 2])), f"C={C}: two-stream forward {rep} differs"
            pipe = model.pipeline(depth=2).prepare(B)
            assert len({id(s) for s in pipe.streams}) == 2
            tickets = [(k % 2, pipe.submit(*batches[k % 2])) for k in range(12)]
            for k, t in tickets:
                assert all(torch.equal(a, b) for a, b in zip(t.result(), serial[k])), f"C={C}: pipelined forward differs"
            pipe.synchronize()
        finally:
            model.set_concurrency(True)
            model.set_gemm_mode(None)
        del model, pipe, serial, tickets
        torch.cuda.empty_cache()


def test_pipeline_follows_reloaded_weights():
    """A pipeline created before load_state_dict must serve the NEW weights afterwards (lanes share the model's packed copy)."""
    from pmce_amd import synth
    J = 17
    model = get_model(J, 256)
    p, f = (T(a).to(dev()) for a in synth.make_inputs(3, J, 31))
    pipe = model.pipeline(depth=2)
    a0 = pipe.submit(p, f).result()[0].clone()
    sd2 = synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=777)
    model.load_state_dict(sd2)
    want = model.forward_with_joints(p, f)[0]
    got = [pipe.submit(p, f).result()[0] for _ in range(2)]        # both lanes
    model.load_state_dict(cached_state_dict(J, 256))                # the model object is shared by the tests of this file
    assert not torch.equal(a0, want)
    assert torch.equal(got[0], want) and torch.equal(got[1], want)


@pytest.mark.parametrize("J,C", [(19, 256), (17, 512)])
def test_streaming_frame_reuse_other_configs(J, C):
    """frame reuse == independent windows for the COCO-19 input and the 512-wide lifter as well."""
    from pmce_amd import streaming, synth
    model = get_model(J, C)
    L = 16 * 3
    p_np, f_np = synth.make_inputs(3, J, 21)
    pose_fr = T(p_np.reshape(-1, J, 2)[:L]).to(dev())
    feat_fr = T(f_np.reshape(-1, 2048)[:L]).to(dev())
    win = streaming.demo_window_list(L)
    ref = streaming.stream_forward(model, pose_fr, feat_fr, windows=win, batch=20, with_joints=True)
    cache = streaming.precompute_frames(model, pose_fr, feat_fr)
    out = streaming.stream_forward_cached(model, cache, windows=win, batch=20, with_joints=True, lanes=2)
    e = [maxabs(a, b) for a, b in zip(out, ref)]
    assert e[0] < 1e-5 and e[1] < 1e-5 and e[2] < 1e-2 and e[3] < 1e-2, e


@pytest.mark.parametrize("C", [256, 512])
def test_gemm_modes(C):
    """The two arithmetic modes of the large products on one batch: (1) they agree to fp32 rounding; (2) the three-product
    f16 form is batch-invariant BITWISE (its k order does not depend on the tile shape the batch size selects) and is what a
    split_f16 model runs at every batch size by default; (3) with a min_batch threshold, batches below it take the fp32 pipe
    (identical to an f32-mode model)."""
    from pmce_amd import synth
    J, B = 17, 64
    model = get_model(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 321)
    p, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    try:
        model.set_gemm_mode("f32")
        assert model.gemm_mode() == "f32"
        o32 = [t.clone() for t in model.forward_with_joints(p, f)]
        o32_small = [t.clone() for t in model.forward_with_joints(p[:4], f[:4])]
        model.set_gemm_mode("split_f16")
        assert model.gemm_mode() == "split_f16"
        osp = [t.clone() for t in model.forward_with_joints(p, f)]
        osp_forced = [t.clone() for t in model.forward_with_joints(p[:4], f[:4])]        # default min_batch = 1: f16 form
        model.set_gemm_mode("split_f16", min_batch=48)
        osp_small = [t.clone() for t in model.forward_with_joints(p[:4], f[:4])]         # below min_batch: fp32 pipe
    finally:
        model.set_gemm_mode(None)
    e = [maxabs(a, b) for a, b in zip(o32, osp)]
    print(f"C={C} f32 vs split_f16 at B=64: mesh {e[0]:.2e} m, pose {e[1]:.2e} m, pose3d {e[2]:.2e} mm, pred {e[3]:.2e} mm")
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM and e[3] < 1000 * TIGHT_M
    for a, b in zip(o32_small, osp_small):
        assert torch.equal(a, b)
    for a, b in zip(osp_forced, osp):
        assert torch.equal(a, b[:4])


@pytest.mark.parametrize("C", [256, 512])
def test_full_forward_batch64_vs_oracle(C):
    """BASELINE configs[1]'s batch through the FULL path, compared with the oracle on every 8th clip (the oracle does 8 clips
    in about a second); C = 512 is north_star's width.  Clips are independent, so the oracle runs the 8 sampled clips as
    one batch of their own."""
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    J, B = 17, 64
    model = get_model(J, C)
    sd = cached_state_dict(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 2024)
    mesh, pose, pose3d, pred = model.forward_with_joints(T(pose2d).to(dev()), T(img_feat).to(dev()))
    idx = list(range(0, B, 8))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d[idx]), T(img_feat[idx]), model.vj_relation)
    e = (maxabs(mesh[idx], rm), maxabs(pose[idx], rp), maxabs(pose3d[idx], rl))
    print(f"B=64 C={C} full forward vs oracle (every 8th clip): mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM


def test_decoder_only_batch64_vs_oracle():
    """BASELINE configs[1] literally: CoEvoDecoder-only forward at batch 64, HIP kernels vs the CPU reference restatement
    (every 8th clip)."""
    from oracle import pmce_oracle as O
    from pmce_amd import models, synth
    J, B = 17, 64
    sd = cached_state_dict(J, 256)
    dec = models.CoevoDecoder.get_model(J, 256)
    dec.load_state_dict({k[len("pose_mesh_coevo."):]: v for k, v in sd.items() if k.startswith("pose_mesh_coevo.")})
    dec = dec.to(dev())
    joints, feats = synth.make_decoder_inputs(B, J, 64)
    pose, mesh = dec(T(joints).to(dev()), T(feats).to(dev()))
    idx = list(range(0, B, 8))
    with torch.no_grad():
        rj, rm = O.decoder_forward(sd, T(joints[idx]), T(feats[idx]), dec.vj_relation, "pose_mesh_coevo.")
    e = (maxabs(pose[idx], rj), maxabs(mesh[idx], rm))
    print("decoder-only B=64 vs oracle (every 8th clip): pose %.2e m, mesh %.2e m" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M


@pytest.mark.parametrize("J,L", [(17, 40), (19, 64)])
def test_streaming_matches_oracle(J, L):
    """SURVEY 8f rank 2 pinned to the ORACLE (not to the HIP forward): a sequence of L frames served through the frame-reuse
    path - one window per frame, the demo's list with its repeated-frame head/tail windows (lib/utils/_dataset_demo.py:91-104;
    a window predicts its middle frame, lib/_img_utils.py:27-55, CoevoDecoder.py:229) - must equal the oracle forward of the
    same windows assembled on the host, and the acceleration error of the streamed middle-frame joints must equal
    oracle/metrics_oracle.compute_error_accel (coord_utils.py:218-245)."""
    from oracle import metrics_oracle as MO
    from oracle import pmce_oracle as O
    from pmce_amd import assets, streaming, synth
    from pmce_amd.eval import Evaluator, H36M_EVAL_JOINT
    model = get_model(J, 256)
    sd = cached_state_dict(J, 256)
    nclip = (L + 15) // 16
    p_np, f_np = synth.make_inputs(nclip, J, 90 + J)
    pose_fr, feat_fr = p_np.reshape(-1, J, 2)[:L], f_np.reshape(-1, 2048)[:L]
    win = streaming.demo_window_list(L)
    assert len(win) == L and (win[:8, 0] == win[:8, 1]).all() and (win[-7:, 0] == win[-7:, 1]).all()
    # host-side assembly of the windows exactly as the reference's get_sequence does it
    wp = np.stack([pose_fr[[s] * 16] if s == e else pose_fr[s:e + 1] for s, e in win])
    wf = np.stack([feat_fr[[s] * 16] if s == e else feat_fr[s:e + 1] for s, e in win])
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(wp), T(wf), model.vj_relation)
    cache = streaming.precompute_frames(model, T(pose_fr).to(dev()), T(feat_fr).to(dev()))
    mesh, pose, pose3d, pred = streaming.stream_forward_cached(model, cache, windows=win, batch=24, with_joints=True, lanes=2)
    e = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    print(f"streamed J={J} L={L} vs oracle: mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM
    # acceleration error of the streamed predictions against a synthetic ground truth, device vs oracle
    jr = torch.from_numpy(assets.load_j_regressor("h36m").astype(np.float32))
    ref_j = torch.einsum("jv,bvl->bjl", jr.double(), rm.double() * 1000)
    ref_j = (ref_j - ref_j[:, :1])[:, list(H36M_EVAL_JOINT)].numpy()
    gt_j = ref_j + np.random.default_rng(0).standard_normal(ref_j.shape) * 5.0
    acc_ref = MO.compute_error_accel(joints_gt=gt_j, joints_pred=ref_j)                     # [L-2]
    pj = pred - pred[:, :1]
    ev = Evaluator(dev())
    acc = ev.accel(pj[:, list(H36M_EVAL_JOINT)].contiguous(), torch.from_numpy(gt_j.astype(np.float32)).to(dev()), np.zeros(L, np.int64))
    ea = float(np.abs(acc.cpu().numpy()[1:-1] - acc_ref).max())
    print(f"   acceleration error of the {L - 2} inner frames: max diff {ea:.2e} mm/frame^2 (mean {acc_ref.mean():.3f})")
    assert ea < 5e-3 and float(acc[0]) == 0.0 and float(acc[-1]) == 0.0


@pytest.mark.parametrize("B", [1, 8])
def test_graphed_forward_matches_eager(B):
    """models.PMCE.graphed(B): the small-batch forward replayed from a hipGraph is bit-identical to the eager call, also
    after the inputs change, and refuses to run on stale weights."""
    from pmce_amd import _lib, synth
    J = 17
    model = get_model(J, 256)
    gf = model.graphed(B)
    for seed in (5, 6):
        p, f = (T(a).to(dev()) for a in synth.make_inputs(B, J, seed))
        ref = [t.clone() for t in model.forward_with_joints(p, f)]
        got = gf(p, f)
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    with pytest.raises(ValueError):
        gf(torch.zeros(B + 1, 16, J, 2, device=dev()), torch.zeros(B + 1, 16, 2048, device=dev()))
    model.load_state_dict(cached_state_dict(J, 256))          # re-pack
    with pytest.raises(_lib.PmceError):
        gf(p, f)


@pytest.mark.parametrize("J,C", [(24, 256), (32, 512)])
def test_other_joint_counts_vs_oracle(J, C):
    """Joint counts beyond the two the reference ships (17, 19): J = 24 / 32 take the generic paths - the one-query
    sequence-attention kernel (N not in {16,17,19}), and the two-launch CrossAttentionBlock (one clip's folded operands no
    longer fit beside the FFN weights for J > 23) - and must still match the oracle.  vj_relation keeps indexing the first 17."""
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    model = get_model(J, C)
    sd = cached_state_dict(J, C)
    pose2d, img_feat = synth.make_inputs(3, J, 314)
    mesh, pose, pose3d = model(T(pose2d).to(dev()), T(img_feat).to(dev()))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation)
    e = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    print(f"J={J} C={C} vs oracle: mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM


def test_entry_points_in_one_process():
    """__graft_entry__.build() (which dlopens libpmce_hip.so) followed by smoke() in ONE fresh process: the library must end up
    in the same HIP runtime as PyTorch whatever the import order (loaded before torch it used to bring in the system
    libamdhip64 as a second runtime that saw no device)."""
    import os.path as osp
    import subprocess
    import sys
    repo = osp.dirname(osp.dirname(osp.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=repo,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "smoke: max-abs vs oracle" in r.stdout


def test_any_finite_feature_magnitude_matches_oracle():
    """The reference accepts any fp32 image feature (raw img_feat enters imgfeat_embed, PoseEstimation.py:80, and the GRU's input
    projection, CoevoDecoder.py:228, un-normalised).  Clips whose features are scaled by 1e5 (beyond f16's 65504), 1e-7 (below
    f16's normal range) and 3e3, and a clip mixing both inside one frame, against the oracle in fp64 - in BOTH product modes: the
    default split-f16 form must be as close as the fp32 pipe (<= 1.5 x its error), and nothing may trip the overflow word
    (a drop-in may not have an error mode the reference lacks)."""
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    J, C, B = 17, 256, 6
    model = get_model(J, C)
    sd = cached_state_dict(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 77)
    img_feat = img_feat.copy()
    img_feat[1] *= 1e5
    img_feat[2] *= 1e-7
    img_feat[3] *= 3e3
    img_feat[4, :, ::2] *= 1e5
    img_feat[4, :, 1::2] *= 1e-7
    img_feat[5, 3] *= 1e5                                        # one frame of a clip only
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation, dtype=torch.float64)
    p2, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    errs = {}
    try:
        for mode in ("f32", "split_f16"):
            model.set_gemm_mode(mode, min_batch=1)
            mesh, pose, pose3d = model(p2, f)
            assert not model.overflowed(), mode                  # (synchronises)
            assert all(torch.isfinite(t).all() for t in (mesh, pose, pose3d)), mode
            errs[mode] = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    finally:
        model.set_gemm_mode(None)
    print(f"features x 1e5 / 1e-7 / mixed vs fp64 oracle: fp32 pipe mesh {errs['f32'][0]:.2e} m pose3d {errs['f32'][2]:.2e} mm; "
          f"split-f16 mesh {errs['split_f16'][0]:.2e} m pose3d {errs['split_f16'][2]:.2e} mm")
    for k, floor, tol in ((0, 2e-6, TIGHT_M), (1, 2e-6, TIGHT_M), (2, 4e-4, TOL_MM)):
        assert errs["split_f16"][k] <= 1.5 * errs["f32"][k] + floor, (k, errs)
        assert errs["split_f16"][k] < tol and errs["f32"][k] < tol, (k, errs)


def test_nonfinite_inputs_propagate_like_the_reference():
    """inf / nan in a clip's features: that clip's outputs are not finite - as the reference's would be - the other clips of the
    batch are bit-identical to a clean run, the model's overflow word REPORTS it and nothing is refused: the next forward runs and
    is bit-identical.  The strict policy refuses until cleared; ``forward_checked`` and ``Pipeline.synchronize`` poll the word
    (nothing did) and name / re-run the batch."""
    from pmce_amd import _lib, synth
    J, C, B = 17, 256, 4
    model = get_model(J, C)
    model.set_gemm_mode("split_f16", min_batch=1)
    pose2d, img_feat = synth.make_inputs(B, J, 31)
    p2, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    try:
        model.set_overflow_policy("report")                         # fully asynchronous calls; the word only reports
        good = [t.clone() for t in model(p2, f)]
        assert not model.overflowed()
        f_bad = f.clone()
        f_bad[1, 5, 77] = float("inf")
        f_bad[3, 0, 5] = float("nan")
        out = [t.clone() for t in model(p2, f_bad)]
        assert model.overflowed()                                   # (synchronises) reported ...
        for k in (1, 3):
            assert not torch.isfinite(out[0][k]).all() and not torch.isfinite(out[2][k]).all()
        for k in (0, 2):                                            # ... neighbours untouched ...
            assert all(torch.equal(o[k], g[k]) for o, g in zip(out, good))
        again = model(p2, f)                                        # ... and nothing is refused
        torch.cuda.synchronize()
        for a, b in zip(again, good):
            assert torch.equal(a, b)
        model.clear_overflow()
        assert not model.overflowed()
        # strict policy: the round-3 behaviour on request
        model.set_overflow_policy(True)
        model(p2, f_bad)
        assert model.overflowed()
        with pytest.raises(_lib.PmceError, match="strict overflow policy"):
            model(p2, f)
        model.clear_overflow()
        model.set_overflow_policy(False)
        # forward_checked: waits, re-runs on the fp32 pipe, clears the word
        outs, reran = model.forward_checked(p2, f_bad)
        assert reran and not model.overflowed()
        for k in (0, 2):          # the clean clips, now from the fp32 pipe: equal to the split form's within fp32 noise
            assert maxabs(outs[0][k], good[0][k]) < TIGHT_M and maxabs(outs[2][k], good[2][k]) < TOL_MM
        assert not torch.isfinite(outs[0][1]).all()               # non-finite inputs stay non-finite (as in the reference)
        outs, reran = model.forward_checked(p2, f)
        assert not reran and all(torch.equal(o, g) for o, g in zip(outs, good))
        # the pipeline polls when it drains and names the batch (default: it only names it and holds no inputs) ...
        pipe = model.pipeline(2)
        ts = [pipe.submit(p2, x) for x in (f, f_bad, f)]
        assert all(t.inputs is None for t in ts)
        with pytest.warns(UserWarning, match="non-finite"):
            named = pipe.synchronize()
        assert named == [1] and pipe.reran == [] and not model.overflowed()
        with pytest.raises(ValueError, match="created with"):
            pipe.synchronize(on_overflow="rerun")
        # ... a ticket the caller dropped is not kept alive by the pipeline (~0.5 GB of retained batches at B = 256)
        import gc
        import weakref
        t = pipe.submit(p2, f)
        w = weakref.ref(t)
        del t
        gc.collect()
        assert w() is None
        assert pipe.synchronize() == []
        # ... and one created with on_overflow="rerun" keeps the inputs and computes the batch again on the fp32 pipe
        pipe = model.pipeline(2, on_overflow="rerun")
        ts = [pipe.submit(p2, x) for x in (f, f_bad, f)]
        with pytest.warns(UserWarning, match="non-finite"):
            named = pipe.synchronize()
        assert named == [1] and pipe.reran == [1] and not model.overflowed()
        for a, b in zip(ts[2].result(), good):
            assert torch.equal(a, b)
        assert pipe.synchronize() == []
        # the module's DEFAULT policy: forward() itself waits, re-runs, clears - non-finite inputs stay non-finite, as in the reference
        model.set_overflow_policy("rerun")
        n0 = model.overflow_reruns
        out = model(p2, f_bad)
        assert model.overflow_reruns == n0 + 1 and not model.overflowed()
        assert not torch.isfinite(out[0][1]).all() and torch.isfinite(out[0][0]).all()
        out = model(p2, f)
        assert model.overflow_reruns == n0 + 1 and all(torch.equal(o, g) for o, g in zip(out, good[:3]))
    finally:
        model.set_overflow_policy("rerun")
        model.set_gemm_mode(None)
        model.clear_overflow()


def test_forward_never_returns_what_the_reference_would_not():
    """VERDICT r04 #7.  Weights that push ONE hidden unit of a decoder MLP past f16's 65504 (fc1 bias 7e4 -> GELU output 7e4): the
    three-product f16 form turns that into inf, the reference (fp32) does not.  Under the module's default overflow policy the
    plain ``model(pose2d, img_feat)`` of lib/core/base.py:222 returns finite values - the batch is computed again on the fp32 pipe,
    bit-identical to a model in 'f32' mode - and they match the oracle on the same weights; 'report' shows what it guards against."""
    from oracle import pmce_oracle as O
    from pmce_amd import assets, models, synth
    J, C, B = 17, 256, 2
    sd = {k: v.clone() for k, v in cached_state_dict(J, C).items()}
    sd["pose_mesh_coevo.coevoblock3.vertx_SA_FFN.mlp.fc1.bias"][5] = 7.0e4
    _MODELS.clear()
    torch.cuda.empty_cache()
    model = models.PMCE.get_model(J, C, 3)
    model.load_state_dict(sd)
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev())
    model.set_gemm_mode("split_f16", min_batch=1)
    assert model.overflow_policy() == "rerun"
    pose2d, img_feat = synth.make_inputs(B, J, 77)
    p2, f = T(pose2d).to(dev()), T(img_feat).to(dev())
    mesh, pose, pose3d = model(p2, f)
    assert model.overflow_reruns == 1 and not model.overflowed()
    assert all(bool(torch.isfinite(t).all()) for t in (mesh, pose, pose3d))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation)
    e = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    scale = max(1.0, float(rm.abs().max()))      # (the 7e4 hidden unit shifts every vertex by the same large offset: tolerance relative to it)
    print(f"re-run on the fp32 pipe vs oracle: mesh {e[0]:.2e} m (max |mesh| {scale:.1f}), pose {e[1]:.2e} m, pose3d {e[2]:.2e} mm")
    assert e[0] < TOL_M * scale and e[1] < TOL_M and e[2] / 1000 < TOL_M
    mesh4, pose4, pose3d4, pred4 = model.forward_with_joints(p2, f)          # the same policy applies
    assert model.overflow_reruns == 2 and torch.equal(mesh4, mesh) and bool(torch.isfinite(pred4).all())
    model.set_overflow_policy("report")
    bad = model(p2, f)
    assert model.overflowed() and not bool(torch.isfinite(bad[0]).all())     # what the default policy keeps away from the caller
    model.clear_overflow()
    model.set_gemm_mode("f32")
    ref32 = model(p2, f)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref32, (mesh, pose, pose3d)))


def test_adversarial_weights_both_modes_vs_oracle():
    """A checkpoint that is not well conditioned: every weight matrix of the lifter and the large decoder products multiplied
    elementwise by a log-normal factor (sigma = 1), plus a 1e3 x outlier weight and a 1e3 x outlier ROW in three of them.  Both
    arithmetic modes against the oracle (torch CPU fp32) on the same weights: the split-f16 form is held to the fp32 pipe's error."""
    from oracle import pmce_oracle as O
    from pmce_amd import assets, models, synth
    J, C, B = 17, 256, 2
    sd = {k: v.clone() for k, v in cached_state_dict(J, C).items()}
    g = torch.Generator().manual_seed(3)
    hit = 0
    for k, v in sd.items():
        if v.ndim == 2 and min(v.shape) >= 64 and ("pose_lifter" in k or "gru_cur" in k or "linear_cur" in k):
            v *= torch.exp(torch.randn(v.shape, generator=g))
            if k.endswith(("attn.qkv.weight", "mlp.fc1.weight", "weight_ih_l0")) and hit < 6:
                v[v.shape[0] // 3, v.shape[1] // 5] *= 1e3
                v[2 * v.shape[0] // 3] *= 1e3
                hit += 1
    assert hit >= 3
    model = models.PMCE.get_model(J, C, 3)
    model.load_state_dict(sd)
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev())
    pose2d, img_feat = synth.make_inputs(B, J, 5)
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation)
        rm64, rp64, rl64 = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation, dtype=torch.float64)
    err = {}
    for mode in ("split_f16", "f32"):
        model.set_gemm_mode(mode, min_batch=1)
        mesh, pose, pose3d = model(T(pose2d).to(dev()), T(img_feat).to(dev()))
        assert not model.overflowed()
        err[mode] = (maxabs(mesh, rm64), maxabs(pose, rp64), maxabs(pose3d, rl64))
    err["oracle fp32"] = (maxabs(rm, rm64), maxabs(rp, rp64), maxabs(rl, rl64))
    scale = (float(rm64.abs().max()), float(rp64.abs().max()), float(rl64.abs().max()))
    print("adversarial checkpoint, error vs an fp64 oracle (mesh m, pose m, pose3d mm):", {k: tuple(f"{x:.2e}" for x in v) for k, v in err.items()},
          "output scales", tuple(f"{x:.2e}" for x in scale))
    for i in range(3):
        assert err["split_f16"][i] <= 1.5 * max(err["f32"][i], err["oracle fp32"][i]) + 1e-7 * scale[i]


@pytest.mark.parametrize("C", [512, 256])
def test_headline_shape_sampled_clips_vs_oracle(C):
    """BASELINE configs[2] directly: one B = 256 forward in the default mode at north_star's width (C = 512) and at the width every
    reference config ships (C = 256: the LayerNorm-epilogue products and, like C = 512, the two-query-tile attention at their full
    size), 16 of its clips against the oracle."""
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    J, B = 17, 256
    model = get_model(J, C)
    sd = cached_state_dict(J, C)
    pose2d, img_feat = synth.make_inputs(B, J, 2024)
    mesh, pose, pose3d, pred = model.forward_with_joints(T(pose2d).to(dev()), T(img_feat).to(dev()))
    assert not model.overflowed()
    idx = list(range(0, B, 16))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d[idx]), T(img_feat[idx]), model.vj_relation)
    e = (maxabs(mesh[idx], rm), maxabs(pose[idx], rp), maxabs(pose3d[idx], rl))
    print(f"B=256, C={C}, 16 sampled clips vs oracle: mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_other_depths_vs_oracle(depth):
    """get_model(num_joint, embed_dim, depth) takes any depth (reference PoseEstimation.py:31-45, cfg.MODEL.hpe_dep): the library
    accepts 1..8; depths other than the shipped 3 against the oracle (which builds the same number of blocks from the same keys)."""
    from oracle import pmce_oracle as O
    from pmce_amd import assets, models, synth
    J, C, B = 17, 256, 2
    sd = synth.make_state_dict(synth.pmce_spec(J, C, depth), seed=321)
    model = models.PMCE.get_model(J, C, depth)
    model.load_state_dict(sd)
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev())
    pose2d, img_feat = synth.make_inputs(B, J, 8)
    mesh, pose, pose3d = model(T(pose2d).to(dev()), T(img_feat).to(dev()))
    with torch.no_grad():
        rm, rp, rl = O.pmce_forward(sd, T(pose2d), T(img_feat), model.vj_relation, depth=depth)
    e = (maxabs(mesh, rm), maxabs(pose, rp), maxabs(pose3d, rl))
    print(f"depth {depth} vs oracle: mesh %.2e m, pose %.2e m, pose3d %.2e mm" % e)
    assert e[0] < TIGHT_M and e[1] < TIGHT_M and e[2] < TOL_MM


def test_smaller_batch_after_a_larger_one_reuses_the_workspace():
    """B = 130 then B = 128 on one engine.  The self-attention scratch used to make the workspace of 129..132 clips
    SMALLER than that of 128, and the engine regrew by batch count: the second call failed check_ws.  Sizes are monotonic now, the engine
    compares bytes, and the B = 128 result is bit-identical to a B = 128 call on a fresh engine's own workspace."""
    from pmce_amd import synth
    J, C = 17, 256
    model = get_model(J, C)
    model.set_gemm_mode("split_f16", min_batch=1)
    try:
        eng = model._ensure_packed()                    # (set_gemm_mode rebuilds the engine lazily: take the one the calls below use)
        eng.ws = None
        p, f = synth.make_inputs(130, J, 4242)
        p, f = T(p).to(dev()), T(f).to(dev())
        big = [t.clone() for t in model.forward_with_joints(p, f)]
        assert model._engine is eng
        n130 = eng.ws.numel()
        small = [t.clone() for t in model.forward_with_joints(p[:128], f[:128])]       # must not raise PMCE_ERR_WORKSPACE
        assert eng.ws.numel() >= eng.lib.pmce_model_workspace_bytes(eng.handle, 128) and eng.ws.numel() >= n130
        eng.ws = None
        fresh = [t.clone() for t in model.forward_with_joints(p[:128], f[:128])]
        for a, b, c in zip(small, fresh, big):
            assert torch.equal(a, b)
            assert torch.equal(a, c[:128])                                             # and a clip's result does not depend on the batch
    finally:
        model.set_gemm_mode(None)


def test_a_stale_overflow_report_does_not_rerun_the_next_call():
    """the overflow word is sticky and shared with the pipeline lanes.  A word left set by an EARLIER asynchronous call
    ("report" policy; here a clip with a NaN input) must not make the next "rerun"-policy forward of CLEAN clips re-run on the fp32 pipe
    (its numbers would depend on what ran before it), and must not be erased unseen: it stays visible through overflowed() until
    clear_overflow()."""
    from pmce_amd import synth
    J, C, B = 17, 256, 2
    model = get_model(J, C)
    model.set_gemm_mode("split_f16", min_batch=1)
    try:
        p, f = synth.make_inputs(B, J, 31)
        p, f = T(p).to(dev()), T(f).to(dev())
        model.set_overflow_policy("rerun")
        model.clear_overflow()
        clean = [t.clone() for t in model(p, f)]
        n0 = model.overflow_reruns
        fbad = f.clone()
        fbad[0, 3, 7] = float("nan")
        model.set_overflow_policy("report")
        model(p, fbad)
        assert model.overflowed()                                   # the report of the asynchronous call
        model.set_overflow_policy("rerun")
        again = model(p, f)
        assert model.overflow_reruns == n0                          # not re-run: the word was not this call's
        assert all(torch.equal(a, b) for a, b in zip(again, clean))
        assert model.overflowed()                                   # and the earlier report is still there to be read
        model.clear_overflow()
        assert not model.overflowed()
    finally:
        model.set_overflow_policy("rerun")
        model.clear_overflow()
        model.set_gemm_mode(None)


@pytest.mark.parametrize("mode", ["default", "f32"])
def test_forward_batch16_matches_reference_fixture(golden, mode):
    """The HIP path against the REFERENCE's own forward on sixteen clips (tests/golden/make_golden_batch.py): joints and the regressed joints in
    full, the mesh at every 13th vertex, ALL vertices through the per-clip sum and sum of squares - in the default arithmetic (three-product
    f16 form at this batch) and on the fp32 pipe."""
    from pmce_amd import synth
    z = golden("e2e_J17_C256_B16_subsampled.npz")
    J, C, B, step = int(z["J"]), int(z["C"]), int(z["B"]), int(z["vertex_step"])
    model = get_model(J, C)
    model.set_gemm_mode(None if mode == "default" else "f32")
    try:
        assert np.array_equal(model.vj_relation, z["vj_relation"])
        pose2d, img_feat = synth.make_inputs(B, J, int(z["input_seed"]))
        mesh, pose, pose3d, pred = model.forward_with_joints(T(pose2d).to(dev()), T(img_feat).to(dev()))
        torch.cuda.synchronize()
    finally:
        model.set_gemm_mode(None)
    e = dict(mesh=maxabs(mesh[:, ::step], T(z["cam_mesh_sub"])), pose=maxabs(pose, T(z["cam_pose"])), pose3d_mm=maxabs(pose3d, T(z["pose3d"])),
             pred_mm=maxabs(pred, T(z["pred_pose"])))
    m64 = mesh.double().cpu()
    e["mesh_sum"] = float((m64.sum(dim=(1, 2)) - T(z["mesh_sum"])).abs().max())
    e["mesh_sumsq_rel"] = float(((m64 * m64).sum(dim=(1, 2)) / T(z["mesh_sumsq"]) - 1).abs().max())
    print(f"B = 16 vs the reference ({mode}):", {k: f"{v:.2e}" for k, v in e.items()})
    assert e["mesh"] < TIGHT_M and e["pose"] < TIGHT_M and e["pose3d_mm"] < TOL_MM and e["pred_mm"] < 1000 * TIGHT_M
    assert e["mesh_sum"] < 0.05 and e["mesh_sumsq_rel"] < 1e-5
