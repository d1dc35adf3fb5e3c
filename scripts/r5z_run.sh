#!/bin/bash
# round 5, call Z: the round's switchable changes off vs on, one box, interleaved graph replays (tile-count-aware 160 tiles, tg_xq_attn, FF row pitch, boundary convs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5z; mkdir -p $O; cd $R
timeout 2400 python scripts/ab.py --rounds 3 --out $O/ab.json --variant r5_off:TG_T160=0,TG_XQ=0,TG_FF_PAD=0,TG_CONV_IN_MFMA=0,TG_CONV_OUT_MFMA=0,TG_CONV_OUT_GN=0 --variant r5_on 2>&1 | tail -16
