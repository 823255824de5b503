mkdir -p gpurun_out/r5_ab_final; O=gpurun_out/r5_ab_final
AB_ROUNDS=2 bash tools/dev_ab_env.sh "cogaps_amd/csrc/libcogaps_hip_AB_old.so" "cogaps_amd/csrc/libcogaps_hip_AB_new.so" -- > $O/old_vs_new.txt 2>&1
AB_ROUNDS=1 bash tools/dev_ab_env.sh "cogaps_amd/csrc/libcogaps_hip_AB_new.so COGAPS_PERSIST=seq" "cogaps_amd/csrc/libcogaps_hip_AB_xnoopq.so COGAPS_PERSIST=seq" "cogaps_amd/csrc/libcogaps_hip_AB_xnofollow.so COGAPS_PERSIST=seq" "cogaps_amd/csrc/libcogaps_hip_AB_xnop1.so COGAPS_PERSIST=seq" "cogaps_amd/csrc/libcogaps_hip_AB_xall.so COGAPS_PERSIST=seq" -- > $O/seq_variants.txt 2>&1
cat $O/old_vs_new.txt $O/seq_variants.txt
