R=$GRAFT_REPO_ROOT
run() { cp $R/ab/$1.so $R/old-audiosync_amd/libaudiosync_hip.so; echo -n "$1 $ASX_EXP_SIDE: "; python3 $R/bench.py --no-cpu --no-config4 --no-single | python3 $R/tools/brief.py; }
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/keep.so
unset ASX_EXP_SIDE; run pm4
for spec in 0,0,0 96,120,40; do export ASX_EXP_SIDE=$spec; run side_pm4; done
unset ASX_EXP_SIDE; run pm4
export ASX_EXP_SIDE=0,0,0; run side_pm4
cp /tmp/keep.so $R/old-audiosync_amd/libaudiosync_hip.so
