"""What does the eager noise kernel in front of every graph replay cost a step?  (a) noise + replay (the product step), (b) replay only (the noise
buffer left as it is): C1 and the default model at 16000 samples.  python tools/replay_only_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.hparam import hparam as hp
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    dev = torch.device('cuda', 0)
    for case, length in (('bench/c1', 16000), ('bench/c3', 16000), ('bench/c3', 160000)):
        hp.set_hparam_yaml(case)
        store = VariableStore(device=dev, seed=2)
        mel = (torch.rand((1, 1 + length // hp.signal.hop_length, hp.signal.n_mels)) * 2 - 1).to(dev)
        model = IAFVocoder(batch_size=1, length=length, store=store)
        model(None, mel, is_training=False)
        g = GraphedVocoder(model)
        g.mel.copy_(mel)
        for name, step in (('noise + replay', lambda: g(g.mel)), ('replay only   ', lambda: g(g.mel, z=g.z))):
            for rounds in range(2):
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    step()
                torch.cuda.synchronize()
                print('%s %6d  %s  %.4f ms per step' % (case, length, name, (time.perf_counter() - t0) / 100 * 1e3), flush=True)
        g.verify()


if __name__ == '__main__':
    main()
