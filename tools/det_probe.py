import sys, torch
sys.path.insert(0, '.')
from lama_amd import trainers, _lib as L
from oracle import lama_oracle as O
cfg = O.BIG_LAMA
sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
sdg = {'generator.' + k: v for k, v in sd.items()}
batch = O.make_synthetic_batch(4, 256, 256, seed=77)
model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
model.load_state_dict(sdg, strict=True); model.freeze().cuda()
for overlap in (False, True):
    for graph in (False, True):
        model.generator.overlap_streams = overlap
        model.generator._plans.clear()
        model.generator.use_graph = graph
        outs = []
        for i in range(6):
            o = model(dict(image=batch['image'].cuda(), mask=batch['mask'].cuda()))['inpainted']
            torch.cuda.synchronize()
            outs.append(o.clone())
        d = [float((outs[i] - outs[0]).abs().max()) for i in range(1, 6)]
        print('overlap', overlap, 'graph', graph, 'max diffs vs first', d, flush=True)
