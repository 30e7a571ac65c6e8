for s in 0 2 4 8 16 32; do echo "stagger=$s"; HSTU_STAGGER=$s python tools/scan_len.py 200 2>&1 | grep N=; done
