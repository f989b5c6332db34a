"""Position arithmetic at the ABI's size limit (just under 2^31 units): tests/big_text.py -- whole text == four shards, the
tail == the oracle, for AhoCorasick, WholeWord and Longest."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_haystack_at_the_size_limit():
    import torch
    if torch.cuda.get_device_properties(0).total_memory < (40 << 30):
        pytest.skip("needs 40 GB of device memory")
    import big_text
    big_text.main()
