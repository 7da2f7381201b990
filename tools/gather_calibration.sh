#!/bin/bash
# FETCH_SIZE calibration for gather patterns, run ON THE GPU BOX:  bash tools/gather_calibration.sh
# -> gpurun_out/gather_calibration.txt : useful bytes / time of each probe launch (tools/probe/gather_probe.hip) and the raw
#    FETCH_SIZE rocprofv3 reports for the same kernels (mean per dispatch).
export TMPDIR=/tmp
O=gpurun_out/gathercal; mkdir -p $O tools/bin
[ -x tools/bin/gather_probe ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/gather_probe.hip -o tools/bin/gather_probe
tools/bin/gather_probe > $O/timing.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc -o p -- tools/bin/gather_probe > $O/pmc.log 2>&1
python - "$O" <<'PY'
import collections, csv, sys
o = sys.argv[1]
agg = collections.OrderedDict()
for r in csv.DictReader(open(o + "/pmc/p_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
with open("gpurun_out/gather_calibration.txt", "w") as f:
    f.write(open(o + "/timing.txt").read())
    f.write("\nraw FETCH_SIZE per dispatch (KiB counter x 1024); 4 dispatches per probe line, in the order of the lines above:\n")
    for k, v in agg.items():
        f.write("%-28s dispatches %2d  raw FETCH MB: %s\n" % (k, len(v), " ".join("%.1f" % (x * 1024 / 1e6) for x in v)))
print(open("gpurun_out/gather_calibration.txt").read())
PY
rm -rf $O/pmc
