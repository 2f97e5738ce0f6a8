"""rocprofv3 kernel symbol -> the name bench.py's roofline uses for the same launches (bench.CFG_NAMES): one mapping for every PMC script."""
import re


def cfg_name_of(k):
    m = re.search(r'igemm_dma_kernel<unsigned short, (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)', k)
    name = f'igemm_dma_kernel<bf16,{",".join(m.groups())}>' if m else k
    m = re.search(r'igemm_dma_kernel<float, (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), 1>', k)
    if m:
        name = f'igemm_dma_kernel<float,{",".join(m.groups())},x3>'
    if 'pw_single_x3' in k:      # (round 4 averaged the decoder's dynamic_layer launches -- 105 MB each -- into this symbol: 0.76 x algorithmic)
        name = 'pw_single_x3_kernel<16,0,256> (dynamic_layer)' if re.search(r'pw_single_x3_kernel<16, 0, 256>', k) else 'pw_single_x3_kernel (HBM-bound 256 -> 256 / 1024 convs, register-resident split weights)'
    if 'wino_x3' in k:
        name = 'wino_x3w_kernel<NB> + small-grid wino_x3_kernel tiles (3x3 / stride 1 as 1-D Winograd F(2,3); FLOPs booked as the direct convolution)'
    if 'bneck_x3' in k:
        name = 'bneck_x3_kernel (conv2 3x3 + conv3 + next conv1, layer1 / layer2 tails)'
    if 'pw_pair' in k:
        name = 'pw_pair_kernel (conv3 + next conv1, layer1)'
    if 'pw_single' in k and 'pw_single_x3' not in k:
        name = 'pw_single_kernel<16,2,0,128> (dynamic_layer)' if re.search(r'pw_single_kernel<16, 2, 0, 128>', k) else 'pw_single_kernel (HBM-bound 1x1 convs, register-resident weights)'
    return name
