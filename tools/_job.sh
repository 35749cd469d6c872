python tools/il_ab.py --var NTRACER_BOX_R64 --values 0,1 --cases band8 --rounds 6 --steps 20
