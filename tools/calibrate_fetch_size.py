"""Calibrates the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on THIS library's access patterns against byte counts
known by construction ("check the x2 FETCH_SIZE correction against a plain streaming-copy kernel of known
size").  MI355X_MICROARCH.md says: both counters in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B / lane) reads; WRITE_SIZE is exact for 16-byte stores.

  python tools/calibrate_fetch_size.py            # driver: runs itself under rocprofv3 --pmc, prints the table
  python tools/calibrate_fetch_size.py --child    # (what runs under the profiler)

Cases (all far larger than L2 + Infinity Cache, every byte touched once):
  copy     : b = 2 * a over a 2 GiB fp32 tensor (torch's vectorised 16 B / lane elementwise kernel)   reads 2 GiB, writes 2 GiB
  gather   : hcspmm forward, D = 32, a permutation matrix over 12 M rows (every X row gathered     reads 12 M x (128 B row + 16 B tiny
             exactly once by a 16 B / lane load of a 128-byte row, the library's own pattern)      descriptor), writes 12 M x 128 B
"""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]


def child():
    import numpy as np
    import torch
    import hcspmm
    dev = torch.device("cuda:0")
    a = torch.empty(1 << 29, dtype=torch.float32, device=dev).normal_()  # 2 GiB
    b = torch.empty_like(a)
    for _ in range(3):
        torch.mul(a, 2.0, out=b)  # an elementwise kernel (a clone may be served by the copy engine)
    del a, b
    N = 12_000_000
    perm = np.random.default_rng(0).permutation(N).astype(np.int32)
    rp = np.arange(N + 1, dtype=np.int32)
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(perm).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, N, N, (N + 15) // 16, rule=2)  # every window on the sparse-row path
    X = torch.randn(N, 32, device=dev)
    for _ in range(3):
        hcspmm.forward(X, rp_d, col_d, *outs)
    torch.cuda.synchronize()


def counters(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tmp = tempfile.mkdtemp(prefix="hcspmm_cal_")
    got = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, c)
        subprocess.run(["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
                        os.path.abspath(__file__), "--child"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        got.update(counters(d))
    N = 12_000_000
    rows = [("copy (b = 2a, 2 GiB, torch elementwise kernel)", "MulFunctor", float(1 << 31), float(1 << 31)),
            ("gather (hcspmm D=32, permutation over 12 M rows)", "hcspmm::hybrid_plan_kernel", N * (128.0 + 16.0), N * 128.0)]
    print("%-52s %14s %14s %8s | %14s %14s %8s" % ("case", "read bytes", "FETCH_SIZE KiB", "ratio", "written bytes", "WRITE_SIZE KiB", "ratio"))
    for name, kern, rd, wr in rows:
        f = [v for (k, c), v in got.items() if kern in k and c == "FETCH_SIZE"]
        w = [v for (k, c), v in got.items() if kern in k and c == "WRITE_SIZE"]
        if not f or not w:
            print(name, "no counters found for", kern)
            continue
        fb, wb = max(f) * 1024.0, max(w) * 1024.0  # the largest kernel of that name (the clone / the forward)
        print("%-52s %14.4g %14.4g %8.3f | %14.4g %14.4g %8.3f" % (name, rd, fb, rd / fb, wr, wb, wr / wb))
    print("ratio = bytes known by construction / (counter x 1024): the correction factor to apply to the counter")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
