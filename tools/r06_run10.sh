cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/build/variants
for u in 2 4; do bash tools/build_variant.sh ks$u gconv.hip "-DGW_KS_UNROLL=$u" > /dev/null 2>&1; done
ls $V
bash tools/prof_solo.sh 64 64 bf16 base64 64 > /dev/null
for u in 2 4; do bash tools/prof_solo.sh 64 64 bf16 ks${u}_64 64 SED_LIB=$V/libvar_ks$u.so SED_ALLOW_VARIANT=1 > /dev/null; done
bash tools/prof_solo.sh 128 256 bf16 basew 24 > /dev/null
bash tools/prof_solo.sh 128 256 bf16 ks2_w 24 SED_LIB=$V/libvar_ks2.so SED_ALLOW_VARIANT=1 > /dev/null
for t in base64 ks2_64 ks4_64 basew ks2_w; do echo "== $t"; grep "gwgrad\|blk0_bwd<\|blk0_fwd\|bconv<0, [0-9]*, 16\|bglu" gpurun_out/solo_$t.md | head -9; done
