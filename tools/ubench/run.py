import ctypes, os, sys
import torch
torch.zeros(1, device="cuda")
name = "libvalu_rates2.so" if len(sys.argv) > 1 and sys.argv[1] == "2" else "libvalu_rates.so"
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), name))
lib.ubench_main()
