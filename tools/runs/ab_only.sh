cd $GRAFT_REPO_ROOT
tools/ab.sh ${AB_VAR:-noxcd} "${AB_CFGS:-mt-f32 mt-bf16 wide-bf16}" ${AB_STEPS:-1500} ${AB_ROUNDS:-3}
