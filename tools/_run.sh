mkdir -p gpurun_out/r03g
tools/abn.sh "ww:8:65536 ww:8:1048576 ww:12:2097152 tt:4:1048576" game_engine_amd/libge_step.so game_engine_amd/ab/al32.so game_engine_amd/ab/al64.so game_engine_amd/ab/al128.so game_engine_amd/ab/kp.so > gpurun_out/r03g/ab.txt 2>&1
grep -c fuse gpurun_out/r03g/ab.txt
