python tools/exp/exp2.py
FPV_DEBUG_FORCE=1 python tools/exp/exp2.py
FPV_DEBUG_FORCE=0 python tools/exp/exp2.py
