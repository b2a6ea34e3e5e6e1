cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/build/variants
bash tools/build_variant.sh grects grec.hip "-DSED_TS" > /dev/null 2>&1
TS_RAW=1 SED_LIB=$V/libvar_grects.so SED_ALLOW_VARIANT=1 timeout 300 python tools/ts_generic.py grec --C 128 --H 256 --dtype bf16 2>&1 | tail -14
