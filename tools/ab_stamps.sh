#!/bin/bash
LSA_EXTRA_FLAGS="-DLSA_NTT_DIAG_STAMPS" python lattisense_amd/build.py --force > /dev/null 2>&1
python tools/ntt_stamps.py 2>&1 | tail -60
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
