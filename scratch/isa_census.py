"""Static census of the hot kernels' inner loops: compiles csrc/<stem>.hip to gfx950 assembly (device only) and prints, for every
basic block of the named kernels that holds matrix or gather instructions, its instruction mix.  argv: none.
-> profiles/r05_isa_census.txt.  (Counts per LOOP BODY as hipcc lays it out: one 16-row tile for the MLP, one round of four edge
groups for the NNConv's steady loop, ...; dynamic counts per tile: the SQ_INSTS_* tables under profiles/.)"""
import collections, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [("nnconv_eg", "nnconv32_eg_kernelILi16ELi4ELi1E"), ("gin", "gin32_mlp_kernel"), ("gin", "gin32_mlp16_kernel"),
           ("gin", "gin32_aggregate_kernel"), ("bn_merge", "merge_bn1_wide_kernel")]
asm = {}
for stem in sorted({k[0] for k in KERNELS}):
    out = os.path.join(tempfile.gettempdir(), f"census_{stem}.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", "-S", "--cuda-device-only",
                    f"{REPO}/tilingnn_amd/csrc/{stem}.hip", "-o", out], check=True, stderr=subprocess.DEVNULL)
    asm[stem] = open(out).read()
for stem, key in KERNELS:
    src = asm[stem]
    m = re.search(r"^(_ZN4tgnn\w*" + re.escape(key) + r"\w*):", src, re.M)
    if not m:
        print(f"{key}: not found"); continue
    body = src[m.end():]
    body = body[:body.index("s_endpgm")]
    blocks, cur, name = [], [], "entry"
    for l in body.split("\n"):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            blocks.append((name, cur)); name, cur = mm.group(1), []
        else:
            t = l.strip()
            if t and not t.startswith((".", ";")):
                cur.append(t.split()[0])
    blocks.append((name, cur))
    print(f"== {key} ({stem}.hip): {sum(len(b) for _, b in blocks)} instructions in {len(blocks)} blocks")
    mb = [collections.Counter(b) for _, b in blocks if any("mfma" in i for i in b)]
    if mb:
        tot = sum(mb, collections.Counter())
        print(f"  all {len(mb)} blocks with matrix instructions together: {sum(tot.values())} instr: vector "
              f"{sum(v for k, v in tot.items() if k.startswith('v_') and 'mfma' not in k)}  matrix {sum(v for k, v in tot.items() if 'mfma' in k)}  "
              f"memory {sum(v for k, v in tot.items() if k.startswith(('buffer_', 'global_')))}  LDS {sum(v for k, v in tot.items() if k.startswith('ds_'))}")
    for name, ins in blocks:
        c = collections.Counter(ins)
        mfma = sum(v for k, v in c.items() if "mfma" in k)
        vmem = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_")))
        if (mfma < 4 and vmem < 4) or len(ins) < 40:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop")))
        wait = sum(v for k, v in c.items() if k.startswith(("s_waitcnt", "s_nop")))
        scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
        print(f"  {name:10s} {len(ins):4d} instr: vector {valu:3d}  matrix {mfma:2d}  memory {vmem:2d}  LDS {lds:2d}  scalar {salu:3d}  waits/nops {wait:2d}  scratch {scr}")
        top = [f"{k} {v}" for k, v in c.most_common(10) if k.startswith("v_") and "mfma" not in k]
        print("             " + ", ".join(top))
