"""Error distribution of the Winograd kernels against float64, per layer shape (tests/test_gpu_wino.py wino_error_row): prints the table
that profiles/r06_wino_error.txt holds.  Units: u = eps32 * rms(y) -- one fp32 rounding of a typical output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_wino as T
    from face_generator_amd.runtime import get_context
    ctx = get_context(0)
    print("| B H W Cin Cout k up | pass | K | wino mean u | wino p99 u | wino max u | direct mean u | direct p99 u | direct max u | mean ratio | max ratio |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for case in T.ERR_CASES:
        r = T.wino_error_row(ctx, *case)
        for which, K in (("fwd", r["K"]), ("dgrad", r["Kd"])):
            g = lambda n: r["%s_%s" % (which, n)]
            print("| %s | %s | %d | %.2f | %.2f | %.1f | %.2f | %.2f | %.1f | %.2f | %.2f |" % (
                " ".join(str(v) for v in case), which, K, g("wino_mean_u"), g("wino_p99_u"), g("wino_max_u"), g("direct_mean_u"),
                g("direct_p99_u"), g("direct_max_u"), g("wino_mean_u") / g("direct_mean_u"), g("wino_max_u") / g("direct_max_u")))


if __name__ == "__main__":
    main()
