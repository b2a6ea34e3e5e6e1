tools/build_variant.sh ts feat.hip "-DSED_TS" >/dev/null 2>&1
SED_LIB=build/variants/libvar_ts.so SED_ALLOW_VARIANT=1 python tools/ts_feat.py 2>&1 | tail -10
