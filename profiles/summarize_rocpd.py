#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database: per-kernel launch statistics and,
when present, PMC counters per kernel.  Usage: summarize_rocpd.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    out = []
    out.append(f"# rocprofv3 summary of {db}\n")
    out.append("## kernel-trace stats (durations from dispatch start/end timestamps)\n")
    out.append("| kernel | calls | total ms | avg us | min us | max us | vgpr | agpr | lds B | grid | wg |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|")
    q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
                  max(d.end-d.start)/1e3, s.arch_vgpr_count, s.accum_vgpr_count, max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, d.grid_size_x order by 3 desc"""
    for r in c.execute(q):
        name = r[0][:70]
        out.append(f"| `{name}` | {r[1]} | {r[2]:.3f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
    try:
        rows = list(c.execute("""select s.kernel_name || ' grid=' || d.grid_size_x, p.name, count(*), avg(e.value), sum(e.value)
                                 from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                                 join rocpd_kernel_dispatch d on d.event_id = e.event_id
                                 join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                                 group by s.kernel_name, p.name, d.grid_size_x order by 5 desc"""))
    except sqlite3.Error as err:
        rows = []
        out.append(f"\n(no PMC data: {err})")
    if rows:
        out.append("\n## PMC counters per kernel\n")
        out.append("| kernel | counter | dispatches | avg per dispatch | total |")
        out.append("|---|---|---|---|---|")
        for r in rows:
            out.append(f"| `{r[0][:70]}` | {r[1]} | {r[2]} | {r[3]:.6g} | {r[4]:.6g} |")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
