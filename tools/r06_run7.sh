cd $GRAFT_REPO_ROOT
bash tools/tl_one.sh waveform-bf16 sg1 > /dev/null
bash tools/tl_one.sh waveform-bf16 sg0 SED_DEBUG=268435456 > /dev/null
paste -d"|" <(cut -c1-70 gpurun_out/tl_waveform-bf16_sg1.txt) <(cut -c1-70 gpurun_out/tl_waveform-bf16_sg0.txt) | head -60
