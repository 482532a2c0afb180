"""Device encoder rates (ldpc_encode_batch / ldpc_encode_random) on the 5G-shaped schedules.
    python tools/enc_bench.py [code frames]"""
import sys, time, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from ldpc_amd import codes, encoder
dev = torch.device("cuda", 0)
cases = [(sys.argv[1], int(sys.argv[2]))] if len(sys.argv) > 2 else [("nr_bg1_z384", 16384), ("nr_bg1_z384", 65536), ("nr_bg2_z40", 1 << 18), ("nr_bg2_z40", 1 << 20)]
for name, fr in cases:
    code = codes.load_packaged(name)
    enc = encoder.device_encoder(code)
    msg = torch.randint(0, 2, (fr, enc.k), device=dev, dtype=torch.uint8)
    cw = torch.empty((fr, enc.n), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): enc.encode_ptr(msg.data_ptr(), fr, cw.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): enc.encode_ptr(msg.data_ptr(), fr, cw.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, fr, f"{ms:.3f} ms  {fr*enc.k/ms/1e6:.1f} Gbit/s info  {fr*(enc.k+enc.n)/ms/1e6:.1f} GB/s")
    if len(sys.argv) > 2: continue
    for _ in range(2): enc.random_codewords(fr, [1], first_frame=0, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): enc.random_codewords(fr, [1], first_frame=0, device=dev)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(name, fr, f"random: {ms:.3f} ms  {fr*enc.k/ms/1e6:.1f} Gbit/s info")
