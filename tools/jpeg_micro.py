# decode time of the bundled photographs: PIL on the host (+ uint8 upload) vs host Huffman + device reconstruction (roma_amd.preproc.decode_jpeg_device)
import glob, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from roma_amd.preproc import decode_jpeg_device
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "assets", "*.jpg"))):
    data = open(f, "rb").read()
    for _ in range(3):
        decode_jpeg_device(data, "cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        decode_jpeg_device(data, "cuda")
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        torch.from_numpy(np.array(Image.open(f).convert("RGB"), dtype=np.uint8)).to("cuda")
    torch.cuda.synchronize()
    t_pil = (time.perf_counter() - t0) / 10
    im = Image.open(f)
    print(f"{os.path.basename(f):20s} {im.size[0]}x{im.size[1]}  PIL decode + upload {t_pil*1e3:6.2f} ms   host Huffman + device reconstruction {t_dev*1e3:6.2f} ms")
