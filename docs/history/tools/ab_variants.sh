#!/bin/bash
# A/B harness: runs docs/history/tools/quick_tput.py against every anonymous-credit-tokens_amd/libact_<name>.so given as arguments
for v in "$@"; do echo "== $v"; ACT_LIB_PATH=$PWD/anonymous-credit-tokens_amd/libact_$v.so NB=${NB:-262144} timeout 200 python docs/history/tools/quick_tput.py 2>&1 | grep -E "device-transcript|k_spend_bits|rror" | head -2; done
