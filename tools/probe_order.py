import sys, os, subprocess
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
mode = sys.argv[1]
def maps(tag):
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
    print(tag, libs, flush=True)
if mode == 'torch_first':
    import torch
    print('torch sees', torch.cuda.is_available(), torch.cuda.device_count(), flush=True)
    torch.zeros(4, device='cuda:0')
    maps('after torch')
    import fastqpacker_amd as fq
    b, n = fq.compress.encode_block(open('tests/golden/sample.fq','rb').read())
    print('encode ok', n, len(b)); maps('after fq')
    t = torch.arange(100, device='cuda:0').sum().item(); print('torch still ok', t)
else:
    import fastqpacker_amd as fq
    b, n = fq.compress.encode_block(open('tests/golden/sample.fq','rb').read())
    print('encode ok', n, len(b)); maps('after fq')
    import torch
    print('torch sees', torch.cuda.is_available(), flush=True)
    maps('after torch')
