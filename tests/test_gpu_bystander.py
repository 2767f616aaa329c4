"""Does work of OTHER kernels change its results because kernels that issue f16 matrix instructions run beside it on the same chip?
(on the builder's boxes, hand-written packed-fp32 arithmetic with op_sel operands of co-resident waves did; cause unknown.)
Part (1) is a REPORT: self-checking bystander kernels of the diagnostics library next to matrix-pipe work; every fresh MI355X the
suite runs on prints the table into the pytest output, so the claim is confirmed or refuted on hardware the builder never saw.
Part (2) is a GATE: the torch kernels the docs name as safe beside a forward (elementwise a*b+c, fp64 reductions, layer_norm,
softmax, gelu - what pmce_amd/eval.py overlaps with pipeline lanes) must be bit-correct beside forwards in BOTH product modes."""
import ctypes as C

import pytest
import torch

from conftest import cached_state_dict

pytestmark = pytest.mark.gpu


def test_bystander_interference_report():
    from pmce_amd import _lib, assets, models, ops, synth
    from scripts.microbench import diag      # bystander / spinner kernels: the diagnostics library, not the product
    lib = _lib.load()
    dlib = diag.load()
    dev = torch.device("cuda:0")
    s_by, s_mx = torch.cuda.Stream(), torch.cuda.Stream()
    lines = []

    # ---- (1) self-checking bystander kernels (scripts/microbench/csrc/dbg_victims.hip: every wave recomputes one fixed function of its own registers
    # 400 times and counts the iterations that differ from the first) next to matrix-pipe work on the other stream
    tab = ((torch.arange(4096 * 1024 + 32 * 1024, device=dev, dtype=torch.int64) % 8191).float() * 0.5).contiguous()
    sink = torch.zeros(256, device=dev)
    M, N, K = 4096, 3072, 2048
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * K ** -0.5
    b = torch.randn(N, device=dev)
    Wp, ws = ops.pack_split_f16(W)
    outg = torch.empty(M, N, device=dev)
    mx = C.c_void_p(s_mx.cuda_stream)

    def gemm_split():
        for _ in range(3):
            _lib.check(lib.pmce_gemm_nt_split_f16(_lib.ptr(A), _lib.ptr(Wp), _lib.ptr(ws), _lib.ptr(b), None, _lib.ptr(outg), M, N, K, K, N, 0, 0, mx))

    def gemm_f32():
        for _ in range(3):
            _lib.check(lib.pmce_gemm_nt_f32(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(outg), M, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, mx))

    def spin(kind):
        return lambda: diag.check(dlib.pmce_dbg_mfma_spin(kind, _lib.ptr(sink), 512, 20000, mx))

    aggressors = [("nothing", lambda: None), ("fp32 GEMM of this path", gemm_f32), ("split-f16 GEMM of this path", gemm_split),
                  ("bare v_mfma_f32_32x32x16_f16", spin(0)), ("bare v_mfma_f32_16x16x32_f16", spin(1)), ("bare v_mfma_f32_32x32x2_f32", spin(4))]
    victims = {6: "v_pk_fma_f32, op_sel operands", 0: "packed fp32, compiler-chosen", 1: "plain fp32", 3: "fp32 matrix instructions", 7: "global loads"}
    lines.append("(1) lanes (of 262,144 x 6 launches) of a self-checking bystander kernel whose result changed at least once:")
    for label, aggr in aggressors:
        row = {}
        for kind, vname in victims.items():
            bad = torch.zeros(4, dtype=torch.int32, device=dev)
            for _ in range(6):
                aggr()
                diag.check(dlib.pmce_dbg_victim(kind, _lib.ptr(bad), 1024, 400, _lib.ptr(tab), C.c_void_p(s_by.cuda_stream)))
                torch.cuda.synchronize()
            row[vname] = int(bad[0])
        lines.append(f"    next to {label:30s}: {row}")

    # ---- (2) what a caller would actually run: a torch elementwise kernel (a * b + c over 64 M floats) on its own stream while a
    # split-mode forward of the model runs on another; compared bitwise with its own stand-alone result
    J, Cw, B = 17, 512, 256
    model = models.PMCE.get_model(J, Cw, 3)
    model.load_state_dict(cached_state_dict(J, Cw))
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    p2, f = (torch.from_numpy(x).to(dev) for x in synth.make_inputs(B, J, 11))
    n = 64 * 1024 * 1024
    ea, eb, ec = (torch.randn(n, device=dev) for _ in range(3))
    ref = ea * eb + ec
    ref_sum = (ea.double() * eb.double()).sum()     # (a reduction kernel as a second kind of foreign work)
    xs = ea[: 16 * 1024 * 1024].reshape(-1, 512)
    ref_ln = torch.nn.functional.layer_norm(xs, (512,))
    ref_sm = torch.softmax(xs, -1)
    ref_ge = torch.nn.functional.gelu(xs)
    torch.cuda.synchronize()
    gate = []
    for mode in ("f32", "split_f16"):
        model.set_gemm_mode(mode, min_batch=1)
        model(p2, f)
        torch.cuda.synchronize()
        wrong_elems = wrong_trials = wrong_sums = wrong_other = 0
        trials = 12
        for _ in range(trials):
            with torch.cuda.stream(s_mx):
                for _ in range(2):
                    model(p2, f)
            with torch.cuda.stream(s_by):
                outs = [ea * eb + ec for _ in range(6)]
                sums = [(ea.double() * eb.double()).sum() for _ in range(2)]
                others = [(torch.nn.functional.layer_norm(xs, (512,)), ref_ln), (torch.softmax(xs, -1), ref_sm), (torch.nn.functional.gelu(xs), ref_ge)]
            torch.cuda.synchronize()
            wrong_other += sum(int((o != r).sum()) for o, r in others)
            bad = sum(int((o != ref).sum()) for o in outs)
            wrong_elems += bad
            wrong_trials += bad > 0
            wrong_sums += sum(int(not torch.equal(s, ref_sum)) for s in sums)
        lines.append(f"(2) torch `a * b + c` (64 M floats, 6 launches x {trials} trials) beside forwards in mode {mode:9s}: {wrong_elems} wrong elements in "
                     f"{wrong_trials} trials; {wrong_sums} of {2 * trials} fp64 reductions differ; layer_norm / softmax / gelu over 16 M floats: {wrong_other} wrong elements")
        gate.append((mode, wrong_elems, wrong_sums, wrong_other))
    model.set_gemm_mode(None)
    print("\n=== bystander interference report (MI355X, this box) ===")
    for ln in lines:
        print(ln)
    print("=== end of report ===")
    for mode, wrong_elems, wrong_sums, wrong_other in gate:
        assert wrong_elems == 0 and wrong_sums == 0 and wrong_other == 0, \
            f"torch kernels beside forwards in mode {mode}: {wrong_elems} wrong elementwise results, {wrong_sums} wrong reductions, {wrong_other} wrong layer_norm/softmax/gelu elements"
