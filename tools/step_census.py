"""Per-STEP kernel census from two rocprofv3 --kernel-trace --stats runs of the same bench command that differ only in
--steps (e.g. 10 and 30): everything outside the timed steps (model build, weight upload, warm-up, graph capture, the
cpu_baseline leg) cancels in the difference, what is left divided by the step difference is what ONE captured step
launches -- by kernel family, with its time.  Answers "are there torch / runtime kernels inside the step?" (VERDICT r2
item 7) without guessing which launches belong to the step.
Usage: python tools/step_census.py <stats_small.csv> <steps_small> <stats_large.csv> <steps_large> [out.txt]"""
import csv
import re
import sys


def load(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return rows


def family(name):
    if name.startswith(("conv_", "void conv_")) or "hdu" in name or re.match(r"^(void )?[a-z_0-9]+_kernel", name):
        return re.sub(r"^void ", "", name).split("(")[0]
    return "[not ours] " + name.split("(")[0][:90]


def main():
    a, na, b, nb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    ds = nb - na
    out, per = [], {}
    for name in set(a) | set(b):
        ca, ta = a.get(name, (0, 0.0))
        cb, tb = b.get(name, (0, 0.0))
        if cb - ca:
            f = family(name)
            c, t = per.get(f, (0.0, 0.0))
            per[f] = (c + (cb - ca) / ds, t + (tb - ta) / ds)
    tot_c, tot_t = sum(v[0] for v in per.values()), sum(v[1] for v in per.values())
    out.append("per step (difference of the %d- and %d-step runs / %d): %.1f launches, %.3f ms of kernel time" % (nb, na, ds, tot_c, tot_t / 1e6))
    foreign = [(f, v) for f, v in per.items() if f.startswith("[not ours]")]
    out.append("launches per step that are NOT this library's kernels: %.2f (%.4f ms)" % (sum(v[0] for _, v in foreign), sum(v[1] for _, v in foreign) / 1e6))
    out.append("%-110s %10s %10s %8s" % ("kernel", "launches", "us/step", "share"))
    for f, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append("%-110s %10.2f %10.1f %7.2f%%" % (f[:110], c, t / 1e3, 100 * t / tot_t))
    text = "\n".join(out)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
