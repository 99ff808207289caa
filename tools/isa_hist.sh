#!/bin/bash
# Static instruction histogram of one kernel of a csrc/*.hip file (gfx950 ISA from hipcc -S):
#   tools/isa_hist.sh ntt_encode.hip 'k_encode_tilesILi10ELb1E'      -> counts per opcode, VGPRs, scratch, LDS
f=$1; pat=$2; root=$(cd "$(dirname "$0")/.." && pwd); tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DLIG_WITH_IPC_COMM --cuda-device-only -S -o $tmp/k.s $root/ligero-prover_amd/csrc/$f 2>/dev/null
awk -v p="$pat" '$0 ~ "^_ZN3lig[0-9]*" p ".*:" {f=1} f{print} f&&/s_endpgm/{exit}' $tmp/k.s > $tmp/body.s
echo "# $f  $pat  ($(grep -cE '^\s+[vs]_|^\s+ds_|^\s+buffer_|^\s+global_' $tmp/body.s) instructions)"
grep -E "^\s+[vs]_|^\s+ds_|^\s+buffer_|^\s+global_" $tmp/body.s | awk '{print $1}' | sort | uniq -c | sort -rn
grep -A40 "amdhsa_kernel _ZN3lig[0-9]*$pat" $tmp/k.s | grep -E "next_free_vgpr|group_segment_fixed_size|private_segment_fixed_size" | head -3
rm -rf $tmp
