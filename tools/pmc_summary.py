#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch) for profiles/.

usage: tools/pmc_summary.py <counter_collection.csv> [title]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    print(f"# {title}")
    print("# source: rocprofv3 --pmc ... --output-format csv (counter_collection.csv); mean value per dispatch")
    for k, ctrs in sorted(agg.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        g, wg, vg, sg, lds = meta[k]
        print(f"\nkernel: {k[:200]}\n  grid={g} wg={wg} vgpr={vg} sgpr={sg} lds={lds}")
        for c, vals in sorted(ctrs.items()):
            print(f"  {c:28s} dispatches={len(vals):3d} mean={sum(vals)/len(vals):.6g}")


if __name__ == "__main__":
    main()
