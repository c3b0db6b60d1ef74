timeout -k 10 120 python tools/debug_route_f64.py seq_prefix170 route_curriculum_prefix170_routeobs_sequence2 170 2>&1 | grep -E "^done|^step 9 |Kernel Name|aborting" | cut -c1-160
exit 0
