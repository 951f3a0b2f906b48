#!/bin/bash
# Ablation build of the fused tile kernel (taylor_fused.inc, -DPPSCI_ABL_BUILD): the primary config's kernel WITHOUT one of
# its parts (PPSCI_ABL=<single-bit mask> at run time), to see what the kernel's time is made of.
#   bash tools/fused_ablate.sh build      (CPU box)  -> paddlescience_amd/libppsci_hip.abl.so
#   bash tools/fused_ablate.sh run <out>  (GPU box)  -> <out>/ablate.jsonl
cd /root/repo
if [ "$1" = build ]; then
  python tools/build_variant.py abl taylor_fused_static_tanh.hip -DPPSCI_FUSED_ONLY_AC -DPPSCI_ABL_BUILD
else
  O=$2
  mkdir -p $O
  : > $O/ablate.jsonl
  for m in 0 1 2 4 8 16 32 64 128 256 512 0; do
    PPSCI_ABL=$m PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.abl.so timeout 200 python tools/fused_main_time.py 100000 | sed "s/^{/{\"abl\": $m, /" >> $O/ablate.jsonl 2>> $O/ablate.err
  done
  cat $O/ablate.jsonl
fi
