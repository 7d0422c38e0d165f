#!/bin/bash
for abl in 0 1 2; do echo "abl $abl: $(MI_STEM_ABL=$abl timeout -k 5 120 python tools/profile_model.py back 256 2>/dev/null | grep -E "stem")"; done
