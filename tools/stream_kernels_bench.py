"""GB/s of the HBM-bound kernels (HIP events, median of 5): algorithmic bytes = every operand read or written once.
Run on the GPU box: python tools/stream_kernels_bench.py  -> one JSON line per kernel (profiles/r02_stream_kernels.jsonl)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aggforce_amd import LinearMap
from aggforce_amd import _kernels as K

PEAK = 8000.0  # GB/s, MI355X HBM3E


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def report(name, nbytes, ms, note=""):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_GB": round(nbytes / 1e9, 3), "GB_per_s": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / PEAK, 3), "note": note}), flush=True)


def main():
    dev = "cuda"
    T, N, n_cg = 1_000_000, 4096, 256
    x = K.synth_normal(T // 4, N, torch.float64, 1, sigma=1.0)           # 24.6 GB
    report("has_nan_kernel<double>", x.numel() * 8, timed(lambda: K.nan_flag(x)), "one read of (T, N, 3)")
    report("sumsq (aggf_sumsq)", x.numel() * 8, timed(lambda: K.sumsq(x)), "one read")
    idx = torch.arange(n_cg, device=dev, dtype=torch.int32) * (N // n_cg)
    ms = timed(lambda: K.slice_gather(x, idx, torch.float64))
    report("slice_gather_kernel<double,double>", 2 * (T // 4) * n_cg * 3 * 8, ms,
           "useful bytes: 24-byte pieces of 128-byte lines (read + write of (T, n_cg, 3))")
    del x
    # K5 augment: c5 shape
    T5, N5, c5 = 500_000, 2048, 128
    co = K.synth_normal(T5, N5, torch.float32, 2, sigma=0.3, lattice=1.5)
    fo = K.synth_normal(T5, N5, torch.float32, 3, sigma=30.0)
    from aggforce_amd.trajectory import CondNormal, Trajectory, AugmentedTrajectory

    cmap = LinearMap([[i * (N5 // c5)] for i in range(c5)], n_fg_sites=N5)
    tr = Trajectory(coords=co, forces=fo)
    ms = timed(lambda: AugmentedTrajectory.from_trajectory(t=tr, augmenter=CondNormal(var=0.01, premap=cmap, seed=1), kbt=0.6955215), 3)
    report("augment (K5 + slice gather of the means)", (2 * T5 * N5 * 3 + 2 * T5 * (N5 + c5) * 3) * 4, ms,
           "read coords+forces, write augmented coords+forces")
    del co, fo, tr
    # K3c trjdot with a per-frame factor
    Tt, Nt, ct = 20_000, 1024, 64
    pts = K.synth_normal(Tt, Nt, torch.float32, 4, sigma=1.0)
    fac = torch.empty((Tt, ct, Nt), dtype=torch.float32, device=dev).normal_()
    report("trjdot_frames_kernel<float,float,float>", fac.numel() * 4 + pts.numel() * 4 + Tt * ct * 3 * 4,
           timed(lambda: K.trjdot_frames(pts, fac)), "(T, n_cg, N) factor read once")
    del fac
    # K4c feat_contract
    Tf, Nf, nf = 4_000, 512, 1024
    ff = K.synth_normal(Tf, Nf, torch.float32, 5, sigma=1.0)
    feat = torch.empty((Tf, Nf, nf), dtype=torch.float32, device=dev).normal_()
    div = torch.empty((Tf, nf, 3), dtype=torch.float32, device=dev).normal_()
    report("feat_contract_kernel<float,float,float>", feat.numel() * 4 + ff.numel() * 4 + 2 * div.numel() * 4,
           timed(lambda: K.feat_contract(ff, feat, div, 0.7, 1024)), "(T, N, n_feat) features read once")
    coef = torch.empty(nf, dtype=torch.float64, device=dev).normal_()
    w = torch.empty((Tf, 1, Nf), dtype=torch.float64, device=dev)
    report("feat_weights_kernel<float>", feat.numel() * 4 + Tf * Nf * 8, timed(lambda: K.feat_weights(feat, coef, w, 0)), "")
    # K1s at CLN025
    Tc, Nc = 4_000_000, 175
    topo = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g4_cln025.npz"))
    cons = {frozenset(int(v) for v in row if v >= 0) for row in topo["pairs"]}
    from aggforce_amd.qp.qplinear import LinearProblem

    prob = LinearProblem(LinearMap([[int(i)] for i in topo["ca"]], n_fg_sites=Nc), cons, torch.device("cuda", 0))
    fc = K.synth_normal(Tc, Nc, torch.float64, 6, sigma=30.0)
    report("gram_small_kernel<double,double> (CLN025, n_red 97)", fc.numel() * 8, timed(lambda: prob.gram(fc)),
           "HBM / MFMA / LDS co-limited, see DESIGN section 6")
    del fc
    # K1s below the ridge point (fp64: 3 n^2 flop per 24 n bytes = n / 8 flop/B against 12.5 of the machine): the same
    # kernel on systems where the HBM alone is the bound
    for Ns in (32, 64, 96, 128):
        Ts = int(16.8e9 / (Ns * 24))
        fs = K.synth_normal(Ts, Ns, torch.float64, 7, sigma=30.0)
        ps = LinearProblem(LinearMap([[0], [Ns // 2]], n_fg_sites=Ns), None, torch.device("cuda", 0))
        report(f"gram_small_kernel<double,double> ({Ns} atoms unconstrained, {Ts} frames)", fs.numel() * 8,
               timed(lambda: ps.gram(fs)), f"{Ns / 8:.0f} flop/B")
        # K3s: a dense 4-site map applied to the same frames (apply_small_kernel)
        ms4 = LinearMap(np.abs(np.random.default_rng(Ns).standard_normal((4, Ns))) + 0.1)
        report(f"apply_small_kernel<double,double> ({Ns} atoms -> 4 sites, {Ts} frames)", fs.numel() * 8 + Ts * 4 * 24,
               timed(lambda: ms4(fs)), "read (T, N, 3), write (T, 4, 3)")
        del fs


if __name__ == "__main__":
    main()
