import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from oracle import oracle as O
from visual_slam_amd import _lib as L
img = np.asarray(Image.open('tests/golden/kitti02/000002.png'))
for mode, fma in [(1, 0), (1, 1)]:
    c = L.Context(max_w=1241, max_h=376, blur_mode=mode, fma=fma)
    c.orb_detect_compute(img)
    _, _, dump = O.orb_detect_compute(img, O.default_params(blur_mode=mode, fma=fma), debug=True)
    lw, lh = dump['lw'], dump['lh']
    ofs = np.concatenate([[0], np.cumsum(lw.astype(np.int64) * lh)])
    for l in range(8):
        b = c.debug_read(L.VO_DBG_BLUR_LEVEL, 0, l).reshape(lh[l], lw[l]).astype(int)
        o = dump['blurred'][ofs[l]:ofs[l+1]].reshape(lh[l], lw[l]).astype(int)
        ys, xs = np.nonzero(b != o)
        print(mode, fma, l, len(ys), list(zip(ys[:6], xs[:6])), (b-o)[ys[:6], xs[:6]] if len(ys) else '')
    c.close()
