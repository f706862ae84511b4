"""Per-basic-block instruction mix of one kernel in an assembly file: bbstat.py file.s <mangled-name-substring> [min]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
bb = []; cur = dict(name="entry", line=start, valu=0, trans=0, ds=0, vmem=0, salu=0, mfma=0, bar=0, wait=0, scr=0, br=[]); bb.append(cur)
for i in range(start + 1, end):
    t = lines[i].strip()
    m = re.match(r"^\.LBB\d+_(\d+):", t)
    if m:
        cur = dict(name=m.group(1), line=i + 1, valu=0, trans=0, ds=0, vmem=0, salu=0, mfma=0, bar=0, wait=0, scr=0, br=[]); bb.append(cur); continue
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    if op == "s_barrier": cur["bar"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): cur["br"].append(op[2:].replace("cbranch_", "") + ":" + t.split()[-1].split("_")[-1]); cur["salu"] += 1
    elif op.startswith("v_mfma"): cur["mfma"] += 1
    elif op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): cur["trans"] += 1; cur["valu"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith("scratch_"): cur["scr"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): cur["vmem"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
tot = 0
for b in bb:
    tot += b["valu"]
    if b["valu"] + b["ds"] + b["vmem"] + b["mfma"] + b["scr"] >= mn:
        print(b["name"], b["line"], "v", b["valu"], "tr", b["trans"], "ds", b["ds"], "vm", b["vmem"], "scr", b["scr"], "s", b["salu"], "mf", b["mfma"], "bar", b["bar"], "w", b["wait"], b["br"])
print("total valu", tot, "blocks", len(bb))
