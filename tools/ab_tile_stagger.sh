for s in 0 4 8 12 16; do echo "stagger $s"; KP1_FU_STAGGER_US=$s python tools/prof_mlp.py 8192 40 2>/dev/null | grep -o "'mlp_train_tile': ([0-9.]*"; done
