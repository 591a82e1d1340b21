import sys, ctypes as C, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l})
if order == 'libfirst':
    from libultrahdr_dev_amd import api
    lib = api.load()
    print('after lib load', maps())
    import torch
    print('after torch import', maps())
    print('torch avail', torch.cuda.is_available())
    print('count', lib.uhdr_hip_device_count())
else:
    import torch
    print('torch avail', torch.cuda.is_available(), maps())
    from libultrahdr_dev_amd import api
    lib = api.load()
    print('after lib load', maps())
    print('count', lib.uhdr_hip_device_count())
