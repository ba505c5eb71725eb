import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
import vector_store_amd as vs
dev = torch.device("cuda:0")
print(json.dumps(bench.i8_callers_record(vs, dev, int(sys.argv[1]), 768, 10, 200, "lowrank", 24, 1.5)))
