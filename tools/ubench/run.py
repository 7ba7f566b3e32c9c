import ctypes, os, sys
import torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_rates.so"))
lib.ubench_main()
