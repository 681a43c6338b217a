"""GPU occupancy of a step from a rocprofv3 --kernel-trace CSV: union of the kernel intervals (any stream) vs wall span,
per step (steps delimited by step_advance_kernel): python scripts/busy.py <kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    marks = [e for s, e, n in rows if "step_advance_kernel" in n]
    for a, b in zip(marks[:-1], marks[1:]):
        iv = [(s, e) for s, e, n in rows if s >= a and e <= b]
        union, cur_s, cur_e, total = 0, None, None, 0
        for s, e in iv:
            total += e - s
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            union += cur_e - cur_s
        print(f"step span {(b - a) / 1e6:7.3f} ms  busy (union) {union / 1e6:7.3f} ms  idle {(b - a - union) / 1e6:6.3f} ms  "
              f"sum of kernel times {total / 1e6:7.3f} ms  launches {len(iv)}")


if __name__ == "__main__":
    main(sys.argv[1])
