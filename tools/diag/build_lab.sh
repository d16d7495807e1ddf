#!/bin/bash
# builds diag_bin/gemm_lab and leaves the kernel's ISA in /tmp/lab
# (the lab compiles its own copy of the kernel files AND links libvitk.so: the executable's definitions win, so the library's
# dispatcher -- the lab's "old" column -- runs the lab's copy of those kernels too; to compare two versions build two labs)
set -e
cd "$(dirname "$0")/../.."
mkdir -p diag_bin /tmp/lab
hipcc $LABFLAGS -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ivit_amd/csrc tools/diag/gemm_lab.hip vit_amd/csrc/gemm_bf16_t128.hip vit_amd/csrc/gemm_bf16_p8.hip -Lvit_amd -lvitk -Wl,-rpath,'$ORIGIN/../vit_amd' -o diag_bin/gemm_lab -save-temps=obj 2>&1 | grep -E "rror" || true
mv diag_bin/*gfx950* diag_bin/*host* diag_bin/*.hipfb /tmp/lab/ 2>/dev/null || true
K=${LABK:-p8}
S=/tmp/lab/gemm_bf16_$K-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "vgpr_count|vgpr_spill" $S | paste - - 
awk '/^_ZN12_GLOBAL__N_1[0-9]*gemm_bf16_[a-z0-9]*_kernelILi0ELin1EEEvNS_[0-9A-Za-z]*ArgsE:/,/s_endpgm/' $S > /tmp/lab/k0.s
grep -n "s_barrier\|s_waitcnt\|global_load_lds\|v_mfma\|ds_read\|s_cbranch" /tmp/lab/k0.s | awk '{print $1,$2,$3}' | awk 'BEGIN{prev="";n=0} {key=$2" "$3; if($2=="v_mfma_f32_16x16x32_bf16"||$2=="global_load_lds_dwordx4"||$2=="ds_read_b128")key=$2; if(key==prev){n++} else {if(prev!="")print prevline, "x"n; prev=key;n=1;prevline=$0}} END{print prevline,"x"n}' > /tmp/lab/k0.sum
