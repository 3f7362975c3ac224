# tuning aid: mixed-radix engine layouts ("<log2 quads per workgroup>,<segment staged in LDS>") over a few lengths
for lay in ${LAYOUTS:-0,1 0,0}; do echo "layout $lay"; SPYHIP_MIX_LAYOUT=$lay python tools/length_probe.py ${LENGTHS:-4000 5000} 2>&1 | grep "N="; done
