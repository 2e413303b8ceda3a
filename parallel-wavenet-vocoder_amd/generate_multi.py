"""generate_multi(*cases): run generate() for several cases (/root/reference/generate_multi.py:11-21)."""
from __future__ import absolute_import, division, print_function

import sys

from .generate import _fire, generate


def generate_multi(*cases):
    for case in cases:
        generate(case)
        print('case \'{}\' Done.'.format(case))
    print('Done.')


if __name__ == '__main__':
    _fire(generate_multi, sys.argv[1:])
