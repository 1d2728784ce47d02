"""Per-kernel statistics from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -> NAME_results.db)."""
import sqlite3
import sys


def main(path, out=None):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                          "from kernels group by name order by 6 desc"))
    lines = ["name,calls,avg_ns,min_ns,max_ns,total_ns"]
    for r in rows:
        lines.append('"%s",%d,%.0f,%d,%d,%d' % (r[0][:80], r[1], r[2], r[3], r[4], r[5]))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
