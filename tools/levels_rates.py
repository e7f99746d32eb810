"""Rates of the level post-processing kernels, device-resident: null count, optional-column
expansion of int64 / int32 values, row starts.  64 Mi slots, definition levels with long and
short stretches of nulls (30 % null), as the levels configuration of the benchmark set.

    python tools/levels_rates.py [slots]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import misc_rates as M  # noqa: E402
import bench  # noqa: E402

torch, pkg = M.torch, M.pkg
L = pkg.lib()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
    rng = np.random.default_rng(0xC3)
    # half the slots in stretches of mean 4, half in stretches of mean 4096
    lens = np.where(rng.random(n // 8) < 0.999, rng.geometric(1 / 4.0, n // 8), rng.geometric(1 / 4096.0, n // 8))
    lens = lens[: np.searchsorted(np.cumsum(lens), n) + 1]
    flags = (rng.random(lens.size) >= 0.3).astype(np.uint8)
    d = np.repeat(flags, lens)[:n]
    if d.size < n:
        d = np.concatenate([d, np.ones(n - d.size, np.uint8)])
    present = int(d.sum())
    dev = M.DEV
    d_t = torch.from_numpy(d).to(dev)
    ws = torch.empty(int(L.pqd_levels_workspace_size(n)) // 8 + 1, dtype=torch.int64, device=dev)
    info = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = M.stream
    print(f"{n} slots, {present} with a value ({100 * (1 - present / n):.1f} % null)")

    def line(name, launch, alg_bytes, check):
        class W:
            def launch(self, s):
                rc = launch(s)
                assert rc == 0, rc
        wall, kms = bench.time_steps(M.S, torch, W(), stream, 20, 3, None)
        torch.cuda.synchronize()
        check()
        t = float(np.mean(kms)) / 1e3
        print(f"{name:44s} {t * 1e6:9.1f} us  algorithmic {alg_bytes / t / 1e9:8.1f} GB/s  frac {alg_bytes / t / 8e12:.3f}", flush=True)

    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    line("count levels equal (NumNulls / NumRows)",
         lambda s: L.pqd_levels_count_equal_device(M.ctx.handle, d_t.data_ptr(), n, 1, ws.data_ptr(), ws.numel() * 8, cnt.data_ptr(), s),
         n, lambda: (_ for _ in ()).throw(AssertionError(int(cnt.item()))) if int(cnt.item()) != present else None)
    for name, dt, w in (("int64", np.int64, 8), ("int32", np.int32, 4)):
        vals = rng.integers(1, 2**31 - 1, size=present).astype(dt)
        v_t = torch.from_numpy(vals).to(dev)
        dst = torch.empty(n, dtype=torch.int64 if w == 8 else torch.int32, device=dev)
        val = torch.empty(n, dtype=torch.uint8, device=dev)

        def check(dst=dst, val=val, vals=vals):
            got = dst[: 1 << 20].cpu().numpy()
            k = int(d[: 1 << 20].sum())
            want = np.zeros(1 << 20, vals.dtype)
            want[d[: 1 << 20] == 1] = vals[:k]
            assert np.array_equal(got, want) and np.array_equal(val[: 1 << 20].cpu().numpy(), d[: 1 << 20])
            assert info.cpu().numpy().tolist() == [present, 0]
        line(f"optional expand {name} (+ validity bytes)",
             lambda s, v_t=v_t, dst=dst, val=val, w=w: L.pqd_optional_expand_device(
                 M.ctx.handle, d_t.data_ptr(), n, 1, v_t.data_ptr(), present, w, 0, dst.data_ptr(), val.data_ptr(), ws.data_ptr(),
                 ws.numel() * 8, info.data_ptr(), s),
             n + present * w + n * w + n, check)
    rep = np.where(rng.random(n) < 0.25, 0, 1).astype(np.uint8)
    r_t = torch.from_numpy(rep).to(dev)
    rows = torch.empty(n + 1, dtype=torch.int64, device=dev)
    n_rows = int((rep == 0).sum())

    def check_rows():
        assert info.cpu().numpy().tolist() == [n_rows, 0]
        assert np.array_equal(rows[:1000].cpu().numpy(), np.flatnonzero(rep[:100000] == 0)[:1000])
    line("row starts of a repeated column (25 % zeros)",
         lambda s: L.pqd_repeated_row_starts_device(M.ctx.handle, r_t.data_ptr(), n, rows.data_ptr(), n + 1, ws.data_ptr(), ws.numel() * 8,
                                                    info.data_ptr(), s),
         n + 8 * n_rows, check_rows)


if __name__ == "__main__":
    main()
