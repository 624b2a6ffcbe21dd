import importlib, sys, time
import torch
sys.path.insert(0, "/root/repo")
rta = importlib.import_module("ray-tracer_amd")
W, H = 3840, 2160
world, cam, keep = rta.composite_scene(rta.teapot_obj_text(1), W, H)
fcam = rta.flatten_camera(cam)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
single = rta.Scene(world)
want = rta.to_ppm_bytes(single.render(fcam)[0])
single.close()
for devices in ([0], [0, 0]):
    multi = rta.MultiScene(world, devices=devices)
    for chunks in (1, 2, 3, 4, 6):
        for _ in range(3):
            multi.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=chunks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            multi.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=chunks)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        ok = (rgb8.cpu().numpy() == want).all()
        print("devices %s chunks_per_gpu %d: %.3f ms per frame, canvas %s" % (devices, chunks, ms, "ok" if ok else "DIFFERS"), flush=True)
    multi.close()
