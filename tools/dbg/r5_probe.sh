R=$GRAFT_REPO_ROOT; cd $R
cp old-audiosync_amd/libaudiosync_hip.so /tmp/keep.so
for l in r_tm r_sp r_v0 r_v2d r_stag r_base r_scan; do cp ab/$l.so old-audiosync_amd/libaudiosync_hip.so; echo "== $l"; for k in 1 2 3; do python3 tools/dbg/peak_probe2.py 2>&1 | grep "lag dev\|refine list"; done; done
cp /tmp/keep.so old-audiosync_amd/libaudiosync_hip.so
