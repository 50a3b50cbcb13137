"""EXPLORATORY: split-bf16 contraction of one 256 x 256 dense layer over a C384 snapshot against the
float64 product -- error next to the fp32 evaluation's, and fp32-equivalent TFLOP/s next to the fp32 MFMA peak (157.3).
    make -C benchmarks/bf16split   (hipcc)      python benchmarks/bf16split/run.py   (GPU box)
Prints one JSON object; bench.py attaches it as a secondary when the library is built."""
import ctypes, json, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
K = F = 256


def bf16_round(x32):
    """float32 -> (bf16 as uint16, its float32 value), round to nearest even (what v_cvt_pk_bf16_f32 does)."""
    bits = np.ascontiguousarray(x32, dtype=np.float32).view(np.uint32)
    rounded = (bits + np.uint32(0x7FFF) + ((bits >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    b = rounded.astype(np.uint16)
    return b, (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def pack_weights(W, ns):
    """[K][F] float32 -> A-operand order [kstep 16][piece][ftile 8][lane 64][8] of bf16."""
    pieces, r = [], W.astype(np.float32).copy()
    for _ in range(ns):
        b, val = bf16_round(r)
        pieces.append(b)
        r = r - val
    out = np.zeros((16, ns, 8, 64, 8), np.uint16)
    lane = np.arange(64)
    for s in range(16):
        for j in range(8):
            k = s * 16 + (lane // 32) * 8 + j
            for t in range(8):
                f = t * 32 + lane % 32
                for p in range(ns):
                    out[s, p, t, :, j] = pieces[p][k, f]
    return out


def main(n=6 * 384 * 384, reps=8):
    lib = ctypes.CDLL(os.path.join(HERE, "libbf16split.so"))
    lib.bf16split_gemm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    W = (rng.uniform(-1, 1, (K, F)) * np.sqrt(6.0 / (K + F))).astype(np.float32)          # Glorot, as bench.zc_spec
    X = torch.relu(torch.randn((K, n), device=dev, generator=torch.Generator(device=dev).manual_seed(1)))  # hidden activations
    C = torch.empty((F, n), device=dev)
    sample = X[:, :4096].cpu().numpy()
    truth = W.astype(np.float64).T @ sample.astype(np.float64)
    err32 = float(np.max(np.abs((W.T @ sample) - truth)) / np.max(np.abs(truth)))         # numpy float32 (BLAS) evaluation
    res = {"workload": f"one dense layer 256 -> 256 over {n} columns, weights Glorot, activations relu(N(0,1))",
           "fp32_numpy_max_rel_err": err32, "fp32_mfma_peak_tflops": 157.3, "variants": []}
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for ns, label in ((1, "bf16 x1 (plain bf16)"), (2, "bf16 x2 (3 MFMAs per product)"), (3, "bf16 x3 (6 MFMAs per product)")):
        Wp = torch.from_numpy(pack_weights(W, ns).view(np.int16)).to(dev)
        lib.bf16split_gemm(ns, Wp.data_ptr(), X.data_ptr(), C.data_ptr(), n, 1, st)
        torch.cuda.synchronize()
        got = C[:, :4096].cpu().numpy()
        err = float(np.max(np.abs(got - truth)) / np.max(np.abs(truth)))
        for _ in range(3):
            lib.bf16split_gemm(ns, Wp.data_ptr(), X.data_ptr(), C.data_ptr(), n, reps, st)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            lib.bf16split_gemm(ns, Wp.data_ptr(), X.data_ptr(), C.data_ptr(), n, reps, st)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        res["variants"].append({"split": label, "max_rel_err_vs_f64": err, "ms_for_%d_passes" % reps: ms,
                                "fp32_equivalent_tflops": 2.0 * K * F * n * reps / (ms * 1e-3) / 1e12})
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
