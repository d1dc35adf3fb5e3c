#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in 1 2 0 1 2 0; do echo "TG_ATTN_FLAGS=$f $(TG_ATTN_FLAGS=$f python scripts/dev_attn_self.py 30 2>/dev/null | head -1)"; done
bash scripts/dev_env_ab.sh TG_ATTN_FLAGS "1 2 0" 3
