import os, sys, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.dist import broadcast_weights, max_over_ranks
sd = weights.make_rec_state_dict("tiny", seed=1)
ms = broadcast_weights([sd], "cuda:0", dist)
print("broadcast ms", ms, "max_over_ranks", max_over_ranks(1.5, "cuda:0", dist))
dist.barrier(); dist.destroy_process_group(); print("nccl ok")
