#!/bin/bash
# tick time by the chunk length of SPLIT calls (needs tools/bin/tune.so, a -DSS_TUNING build): tools/sweep_tick_lsplit.sh [L ...]
for L in ${@:-30 40 48 50 60}; do echo "== L_split $L"; SS_TD_LSPLIT=$L SOUNDSCOPE_HIP_LIB=tools/bin/tune.so timeout 120 python tools/probe_tick_host.py | grep -E "tick wall|wait|separate"; done
