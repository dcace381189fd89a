for v in 2 4; do echo "LUT variant $v"; VOXCARVE_LUT_VARIANT=$v python scripts/exp_cams.py 1024 lut; done
