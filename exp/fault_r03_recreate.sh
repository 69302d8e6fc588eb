#!/bin/bash
# exp/fault_r03_recreate.sh [OUT]: the build that faulted on the GPU in round 3, recreated on the CPU, and what its
# scratch instructions are.  The faulting source was never committed: revision 1b04880 is the
# fix (the sort / reduce phase of the per-tile backwards as a textual include); this script turns that include back
# into the __forceinline__ device function with a dozen pointer parameters it had been, compiles the unit to
# assembly with the resource remarks, and classifies every scratch_* instruction of the instance that was running
# (grad_fused_kernel<SH, 9, single march>).  That the recreation IS the faulting build is shown by its numbers: 88
# bytes of scratch for that instance, 68-112 for the SH9 instances, 0 for the others -- NOTEBOOK.md 4.1's record of
# the incident -- whereas a plain (not inlined) function gives 340-380.  No GPU needed.
set -e
root=$(cd $(dirname $0)/.. && pwd)
out=${1:-$root/profiles/r04_fault_isa.txt}
tmp=$(mktemp -d)
git -C $root archive 1b04880 svox_t_amd/csrc include | tar -x -C $tmp
cd $tmp/svox_t_amd/csrc
python3 - <<'PY'
inc = open('svoxt_tile_reduce.inc').read()
body = inc[inc.index('\n{\n') + 1:]
body = body.replace('s_nb = incl', '*s_nb = incl').replace('readfirstlane(s_nb)', 'readfirstlane(*s_nb)')
fn = '''
template <int FMT, int BD, int K, int T, int R, int NT, bool COUNT>
__device__ __forceinline__ void tile_sort_reduce(int lane, int wave, int32_t* keys, int32_t* cnt, uint16_t* order, const uint32_t* r_sl,
                                  const float* r_sg, const float* r_w, const float* r_c, const float* bases, const float* gl,
                                  float* stage, int32_t* seg, int32_t* s_nb, float* __restrict__ grad, int gstride,
                                  unsigned long long* __restrict__ counters) {
    constexpr int C = 3;
    constexpr int NB = (FMT == FMT_SH) ? BD : 0;
    constexpr int BDS = (FMT == FMT_SH) ? (BD | 1) : 1;
    constexpr int HALF = K < 16 ? K : 16;
    constexpr int KS = HALF | 1;
'''
src = open('svoxt_bwd_kernels.h').read()
call = '        tile_sort_reduce<FMT, BD, K, T, R, NT, COUNT>(lane, wave, keys, cnt, order, r_sl, r_sg, r_w, r_c, bases, gl, stage, seg, &s_nb, grad, gstride, counters);\n'
assert src.count('#include "svoxt_tile_reduce.inc"\n') == 1
src = src.replace('#include "svoxt_tile_reduce.inc"\n', call)
anchor = '// The backward of an image in ONE kernel after the forward'
src = src.replace(anchor, fn + body + '}\n\n' + anchor)
open('svoxt_bwd_kernels.h', 'w').write(src)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o k.s svoxt_kernels.hip 2> remarks.txt
python3 - "$out" <<'PY'
import re, subprocess, sys
from collections import Counter
txt = open('remarks.txt').read()
cur, res = None, {}
for line in txt.splitlines():
    m = re.search(r'remark: (.*) \[-Rpass', line)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith('Function Name:'):
        cur = s.split(':', 1)[1].strip(); res[cur] = {}
    elif cur and ':' in s:
        k, v = s.rsplit(':', 1); res[cur][k.strip()] = v.strip()
names = [n for n in res if 'grad_fused_kernel' in n]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
out = ["# exp/fault_r03_recreate.sh: revision 1b04880 with the sort / reduce phase as a __forceinline__ function (the r03 faulting build)",
       "# scratch bytes per lane, VGPRs, dynamic stack, instance"]
for n, d in zip(names, dem):
    out.append(f"{res[n].get('ScratchSize [bytes/lane]'):>4} {res[n].get('VGPRs'):>4} {res[n].get('Dynamic Stack'):>6}  {d.split('(')[0]}")
s = open('k.s').read()
m0 = re.search(r'\n(_ZN5svoxt17grad_fused_kernelILi1ELi9ELb0ELb0ELi0\S*):', s)
body = s[m0.start():s.index('.Lfunc_end', m0.start())].splitlines()
sc = [(i, l.strip()) for i, l in enumerate(body) if re.match(r'\s*scratch_', l)]
forms = Counter()
for _, l in sc:
    op, args = re.match(r'(scratch_\w+)\s+([^;]*)', l).groups()
    cls = lambda x: 'VGPR' if re.match(r'v\d|v\[', x) else 'SGPR' if re.match(r's\d|s\[', x) else x.split()[0]
    forms[(op, tuple(cls(a.strip()) for a in args.split(',')), 'Folded Spill' in l or 'Folded Reload' in l)] += 1
out.append("")
out.append(f"# grad_fused_kernel<SH, 9, EXACT = false> (the instance that was running): {len(body)} lines of ISA, {len(sc)} scratch instructions")
out.append("# count  opcode  operands (loads: vdst, vaddr, saddr; stores: vaddr, vdata, saddr; 'off' = no register: every address is flat-scratch base + constant)  marked by the compiler")
for (op, ops, spill), n in sorted(forms.items()):
    out.append(f"{n:>4}  {op}  {ops}  {'spill/reload' if spill else 'NOT a spill'}")
offs = sorted(set(int(m.group(1)) for _, l in sc for m in [re.search(r'offset:(\d+)', l)] if m))
out.append(f"# constant offsets used: {offs} (+ 0): inside the 88-byte frame; no scratch instruction takes an address register")
# the EXEC state at the stores: the last exec-changing instruction in front of each store
execre = re.compile(r's_(and|or|xor|andn2)_saveexec|s_(or|and|andn2|xor|mov)_b64 exec')
for i, l in sc:
    if 'store' not in l: continue
    j = max((k for k in range(i) if execre.search(body[k])), default=None)
    out.append(f"  line {i:>5} {l.split(';')[0].strip():<52} last EXEC change before it: line {j}: {body[j].strip() if j is not None else '-'}")
open(sys.argv[1], 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $tmp
