"""Train and evaluate STOVE:  python run_stove.py --args key value key value ...
(same entry point and argument convention as the reference's run_stove.py:8-14;
multi-GPU: python -m torch.distributed.run --nproc-per-node N run_stove.py --args ...)."""
import sys

from model.main import main

if __name__ == '__main__':
    pairs = sys.argv[2:]
    trainer = main(sh_args=dict(zip(pairs[0::2], pairs[1::2])))
    trainer.train()
