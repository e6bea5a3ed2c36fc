import io, os, sys, time, pickle
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imageflow_amd.codecs import mozjpeg_decoder as D
files = pickle.load(open("/tmp/ifhip_entropy_files.pkl", "rb"))
torch.zeros(1, device="cuda").item()
for rep in range(4):
    t0 = time.perf_counter()
    ent = D.JpegEntropyBatch(files)
    torch.cuda.synchronize()
    print("create ms", round((time.perf_counter() - t0) * 1e3, 2), file=sys.stderr)
    del ent
