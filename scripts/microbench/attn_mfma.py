"""Lifter attention at the headline shape (B = 256, T = 16, J = 17, C = 512 and 256): the matrix-pipe kernel of the split-f16 mode
(seq_attention_mfma.hip) against the vector-pipe kernel (lifter.hip), spatial and temporal, with the HBM bytes each moves.
   python scripts/microbench/attn_mfma.py [B]"""
import sys
import os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from pmce_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
T, J = 16, 17


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for C in (512, 256):
    M = B * T * J
    g = torch.Generator(device="cpu").manual_seed(5)
    qkvs = [torch.randn(M, 3 * C, generator=g).to(dev) for _ in range(3)]     # rotate buffers: 3 x 428 MB > the 256 MB cache
    mbytes = M * 4 * C * 4 / 1e6
    for name, args in (("spatial ", (B * T, J, C, 0, J, 0, 1)), ("temporal", (B * J, T, C, J, 1, T * J, J))):
        i = [0]

        def vec():
            i[0] += 1
            ops.seq_attention(qkvs[i[0] % 3], *args, out_split=True)

        def mat():
            i[0] += 1
            ops.seq_attention_split(qkvs[i[0] % 3], *args)

        tv, tm = timeit(vec), timeit(mat)
        print(f"C={C} B={B} {name}: vector pipe {tv:7.1f} us ({mbytes / tv:5.2f} TB/s)   matrix pipe {tm:7.1f} us ({mbytes / tm:5.2f} TB/s)   [{mbytes:.0f} MB per launch]", flush=True)
