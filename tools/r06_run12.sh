cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/build/variants
bash tools/build_variant.sh u1o3 blk0.hip "-DBLK0_BWD_OCC16=3 -DBLK0_BWD_GUNROLL=1" > /dev/null 2>&1
bash tools/build_variant.sh u2o3 blk0.hip "-DBLK0_BWD_OCC16=3 -DBLK0_BWD_GUNROLL=2" > /dev/null 2>&1
bash tools/build_variant.sh u4o2 blk0.hip "-DBLK0_BWD_OCC16=2 -DBLK0_BWD_GUNROLL=4" > /dev/null 2>&1
bash tools/build_variant.sh u1o2 blk0.hip "-DBLK0_BWD_OCC16=2 -DBLK0_BWD_GUNROLL=1" > /dev/null 2>&1
bash tools/prof_solo.sh 64 64 bf16 d64 64 > /dev/null
for v in u1o3 u2o3 u4o2 u1o2; do bash tools/prof_solo.sh 64 64 bf16 ${v}_64 64 SED_LIB=$V/libvar_$v.so SED_ALLOW_VARIANT=1 > /dev/null; done
for t in d64 u1o3_64 u2o3_64 u4o2_64 u1o2_64; do echo "== $t"; grep "blk0_bwd<" gpurun_out/solo_$t.md | head -3; done
