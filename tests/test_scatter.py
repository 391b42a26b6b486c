"""torch_scatter drop-in (scatter_max / scatter_min) vs the sequential oracle, and the
reference's use of it (majority class per voxel, h3dgsv3.py:270-291)."""
import numpy as np
import pytest
import torch

from oracle import scatter_oracle


def test_oracle_known_answers():
    src = np.array([3, 9, 9, -2, 7, 7], dtype=np.int64)
    idx = np.array([0, 0, 0, 2, 3, 3], dtype=np.int64)
    out, arg = scatter_oracle.scatter_arg(src, idx)
    assert out.tolist() == [9, 0, -2, 7] and arg.tolist() == [1, 6, 3, 4]  # first of ties; empty group -> (0, n)
    out, arg = scatter_oracle.scatter_arg(src, idx, is_min=True)
    assert out.tolist() == [3, 0, -2, 7] and arg.tolist() == [0, 6, 3, 4]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.float32])
@pytest.mark.parametrize("is_min", [False, True])
def test_scatter_arg_matches_oracle(dtype, is_min, dev):
    import artdeco_amd
    artdeco_amd.install_dropins()
    import torch_scatter
    g = torch.Generator().manual_seed(11)
    for n, groups in ((0, 0), (1, 1), (257, 40), (100_003, 5_000)):
        if dtype.is_floating_point:
            src = (torch.randint(-6, 6, (n,), generator=g).float() * 0.5)  # many exact ties, negative values
        else:
            src = torch.randint(-5, 5, (n,), generator=g).to(dtype)
        idx = torch.randint(0, max(groups, 1), (n,), generator=g)
        dim_size = groups + 3 if n else 4  # trailing empty groups
        fn = torch_scatter.scatter_min if is_min else torch_scatter.scatter_max
        out, arg = fn(src.to(dev), idx.to(dev), dim_size=dim_size)
        ro, ra = scatter_oracle.scatter_arg(src.numpy(), idx.numpy(), dim_size, is_min)
        assert out.dtype == dtype and arg.dtype == torch.int64
        assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(arg.cpu().numpy(), ra)


@pytest.mark.gpu
def test_majority_class_per_voxel_like_update_voxel(dev):
    """The exact op sequence of h3dgsv3.py:276-291 around scatter_max, checked against a python vote count."""
    import artdeco_amd
    artdeco_amd.install_dropins()
    from torch_scatter import scatter_max
    g = torch.Generator().manual_seed(3)
    n, n_vox, n_cls = 20_000, 700, 90
    inv_idx = torch.randint(0, n_vox, (n,), generator=g).to(dev)
    cls = torch.randint(0, n_cls, (n,), generator=g).to(dev)
    offset = int(cls.max()) + 1
    pair_unique_ids, pair_counts = torch.unique(inv_idx * offset + cls, return_counts=True)
    v_in_pair, c_in_pair = pair_unique_ids // offset, pair_unique_ids % offset
    _, max_indices = scatter_max(pair_counts, v_in_pair)  # default dim_size, as the reference calls it
    mode = c_in_pair[max_indices].cpu().numpy()
    votes = np.zeros((n_vox, n_cls), dtype=np.int64)
    np.add.at(votes, (inv_idx.cpu().numpy(), cls.cpu().numpy()), 1)
    present = votes.sum(1) > 0
    assert present.all()
    assert np.array_equal(mode, votes.argmax(1))  # first maximum = smallest class id, as sorted pair ids give


@pytest.mark.gpu
def test_scatter_rejects_what_it_does_not_implement(dev):
    import artdeco_amd
    artdeco_amd.install_dropins()
    import torch_scatter
    a = torch.zeros(4, 4, device=dev)
    with pytest.raises(NotImplementedError):
        torch_scatter.scatter_max(a, a.long())
    with pytest.raises(Exception):
        torch_scatter.scatter_max(torch.zeros(4), torch.zeros(4, dtype=torch.long))  # CPU tensors: no fallback
