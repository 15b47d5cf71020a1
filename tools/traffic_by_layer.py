#!/usr/bin/env python3
"""Per-launch-shape breakdown of the fabric traffic of the position-major convs: which layers carry the 1.52 x?
usage: traffic_by_layer.py <dir_fetch> <dir_write>     (directories of the two rocprofv3 --pmc passes, as tools/traffic_json.py takes them)
Groups the dispatches of conv_igemm_f32_kernel<..., POS=true, ...> by (instantiation, grid) in launch order and prints, per group,
launches, fetched MB (FETCH_SIZE KB x 2: gfx950 correction) and written MB per launch."""
import csv, glob, sys, collections


def load(d, counter):
    names, grids, order = {}, {}, {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
            grids[row["Dispatch_Id"]] = (int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1), int(row["Grid_Size_Y"]), int(row["Grid_Size_Z"]))
            order[row["Dispatch_Id"]] = int(row["Start_Timestamp"])
    vals = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    return names, grids, order, vals


nf, gf, of, vf = load(sys.argv[1], "FETCH_SIZE")
nw, gw, ow_, vw = load(sys.argv[2], "WRITE_SIZE")


def seq(names, grids, order, vals):
    ids = [i for i in sorted(names, key=lambda i: order[i]) if "conv_igemm_f32_kernel" in names[i] and ", true, " in names[i]]
    return [(names[i].split("conv_igemm_f32_kernel")[1].split("(")[0], grids[i], vals.get(i, 0.0)) for i in ids]


sf, sw = seq(nf, gf, of, vf), seq(nw, gw, ow_, vw)
assert len(sf) == len(sw), (len(sf), len(sw))
groups = collections.OrderedDict()
for (a, g, f), (b, g2, w) in zip(sf, sw):
    assert a == b and g == g2
    e = groups.setdefault((a, g), [0, 0.0, 0.0])
    e[0] += 1; e[1] += 2 * f * 1024 / 1e6; e[2] += w * 1024 / 1e6
tf = tw = 0.0
print(f"{'instantiation':34s} {'workgroups':>16s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s}")
for (a, g), (n, f, w) in groups.items():
    print(f"{a:34s} {str(g):>16s} {n:8d} {f / n:16.1f} {w / n:16.1f}")
    tf += f; tw += w
print("total fetch MB", round(tf, 1), "write MB", round(tw, 1), "dispatches", len(sf))
