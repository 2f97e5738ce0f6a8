# runtime settings of this repo's configs: inference only; one process per GPU over RCCL
dist_params = dict(backend='nccl')
log_level = 'INFO'
load_from = None
