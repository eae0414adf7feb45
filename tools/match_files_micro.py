# end-to-end rate of match() on JPEG FILE PATHS (decode + upload + resize + normalise + the full 560 -> 864 match), device JPEG path vs PIL
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "assets")
a, b = os.path.join(root, "sacre_coeur_A.jpg"), os.path.join(root, "sacre_coeur_B.jpg")
model = bench.build_model("cuda", torch.float16)
for flag in (True, False):
    model.device_jpeg = flag
    for _ in range(3):
        model.match(a, b, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model.match(a, b, device="cuda")
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 20
    print(f"match(path, path), 640x480 + 618x640 JPEGs, {'host Huffman + device reconstruction' if flag else 'PIL decode on the host':40s}: {t*1e3:6.2f} ms per pair = {1/t:5.1f} pairs/s")
