python tools/repro_c4_aug_capture.py > gpurun_out/r5f_repro_capture.txt 2>&1
python tools/profile_c3_api_host.py 100 > gpurun_out/r5f_profile_c3_api_host.txt 2>&1
