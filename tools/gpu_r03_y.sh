cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_y; mkdir -p $O; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o x --output-format csv -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/prof.err
f=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $O/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:60] for r in rows]
# last step = last third
n = len(names) // 3
last = names[-n:]
import collections
prev = collections.Counter()
for i, k in enumerate(last):
    if "copyBuffer" in k:
        prev[(last[i - 1] if i else "-", last[i + 1] if i + 1 < len(last) else "-")] += 1
for (a, b), c in prev.most_common(30):
    print(c, "|", a, "| -> copyBuffer -> |", b)
print("launches in last step:", n)
PY
rm -rf $O/prof; cat $O/seq.txt
