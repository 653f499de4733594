#!/usr/bin/env python
"""Per-kernel lint of a gfx950 assembly listing for the patterns that cost r05 its largest gains:
`s_waitcnt vmcnt(0)` inside loops, scratch (spill) accesses inside loops, single `v_cvt_pk_bf16_f32 x, 0` conversions
(+ `v_perm_b32` merges), `v_max x, x, x` canonicalisations, and accumulators shuttled through `v_accvgpr_read / _write / _mov`
inside loops (MFMAs in the AGPR form with a destination different from their accumulator source: launch bounds without a
minimum workgroup count — the weight-gradient kernel moved all 72 of its accumulators every step).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Irl4co_amd/csrc -S --cuda-device-only \
          -o /tmp/k.s rl4co_amd/csrc/am_attn_flash.hip
    python tools/isa_lint.py /tmp/k.s [kernel-name substring]
"""
import re,sys,collections,subprocess
t=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else ''
for f in re.split(r'\n(?=_Z[\w]+:)', t):
    name=f.split(':',1)[0]
    if not name.startswith('_Z') or pat not in name: continue
    L=f.split('\n')
    labels={}
    for i,l in enumerate(L):
        m=re.match(r'(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=i
    loops=[]
    for i,l in enumerate(L):
        m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)',l)
        if m:
            tgt=m.group(1) or m.group(2)
            if tgt in labels and labels[tgt]<i: loops.append((labels[tgt],i))
    inl=lambda i: any(a<=i<=b for a,b in loops)
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','')[:70]
    vm0=sum(1 for i,l in enumerate(L) if 's_waitcnt vmcnt(0)' in l and inl(i))
    scr=sum(1 for i,l in enumerate(L) if 'scratch_' in l and inl(i))
    single=sum(1 for l in L if re.search(r'v_cvt_pk_bf16_f32 v\d+, v\d+, s\d+',l))
    canon=sum(1 for l in L if re.search(r'v_max_f32_e32 (v\d+), (v\d+), \2\b',l))
    agpr=sum(1 for i,l in enumerate(L) if 'v_accvgpr_' in l and inl(i))
    # (r06) agent-scope fences: buffer_wbl2 is an L2 write-back (0.6 - 1.0 ms of the teacher launch per fence and workgroup)
    wbl2=sum(1 for l in L if 'buffer_wbl2' in l or 'buffer_inv' in l)
    # (r06) a vmcnt wait that can only be for a load issued a few instructions earlier IN a loop (an exposed round trip):
    # s_waitcnt vmcnt(0..1) within 10 instructions behind a global / buffer load, no store in between
    early=0
    for i,l in enumerate(L):
        if inl(i) and re.search(r's_waitcnt vmcnt\([01]\)',l):
            back=[x for x in L[max(0,i-10):i] if not x.strip().startswith(';')]
            if any(re.search(r'(global|buffer|flat)_load',x) for x in back): early+=1
    print(f"{dn:70s} lines {len(L):5d} loop-vmcnt0 {vm0:3d} loop-scratch {scr:3d} single-cvt {single:4d} canon {canon:3d} loop-agpr-moves {agpr:3d} wbl2/inv {wbl2:2d} early-wait {early:3d}")
