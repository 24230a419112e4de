"""Entry point mirroring the reference's test_modelnet_AE_dr.py: identical to test_modelnet_AE.py except for the
weights directory it evaluates (the model trained with latent dropout); pass it with --load-path."""
import runpy
import sys

if __name__ == '__main__':
    sys.argv[0] = 'test_modelnet_AE.py'
    runpy.run_module('test_modelnet_AE', run_name='__main__')
