cd ${GRAFT_REPO_ROOT:-/root/repo}
for R in 1 2; do for V in libsdr_amd/libsdrhip_nowl.so libsdr_amd/libsdrhip.so; do echo "== $V"; SDRHIP_LIB=$V timeout 200 python tools/probes/bigconv_time.py 2>&1 | grep "FilterNode<float>" | grep -E "\( *(512|1024|2048|4096|1000|6000|8192|16384|12000)\)" | cut -c1-140; done; done
