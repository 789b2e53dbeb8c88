import re,sys
for f in sys.argv[1:]:
    lines=open(f).read().split("\n")
    depth=0; cur={}
    tot={}
    name=None
    for l in lines:
        m=re.match(r"^(\.LBB\d+_\d+):",l)
        if m: depth=0; name=m.group(1)
        m2=re.search(r"Depth=(\d+)",l)
        if l.startswith(";") and m2: depth=max(depth,int(m2.group(1)))
        m3=re.match(r"^\s+([a-z_0-9]+)",l)
        if m3:
            op=m3.group(1)
            k=None
            if re.match(r"v_readlane|v_readfirstlane",op): k="rdl"
            elif op.startswith("v_writelane"): k="wrl"
            elif op.startswith("scratch_"): k="scratch"
            elif op.startswith("v_"): k="valu"
            elif op.startswith("s_"): k="salu"
            if k: tot.setdefault(depth,{}).setdefault(k,0); tot[depth][k]+=1
    print(f)
    for d in sorted(tot): print("  depth",d,tot[d])
