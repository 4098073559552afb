"""Table format and preparation: irbfn_amd.tables against literal restatements of the
statements in scripts/train_nmpc.py:48-184 and scripts/train_nmpc_frenet.py:48-212.  No reference table
survives in the repository (.MISSING_LARGE_BLOBS) -> synthetic tables on the reference's grid layout."""
import numpy as np
import pytest

from irbfn_amd import tables
from irbfn_amd.model import WCRBFNet


def _grid_table(rng, kind, T=5, infeasible=0.0):
    if kind == tables.CARTESIAN:
        axes = [np.linspace(0.5, 7.0, 4), np.linspace(0.0, 3.6, 5), np.linspace(0.0, 3.6, 5), np.linspace(0.0, 3.1, 4),
                np.linspace(0.5, 7.0, 3), np.linspace(-0.4, 0.4, 3), np.linspace(-2.0, 2.0, 3)]
    else:
        axes = [np.linspace(0.0, 0.8, 4), np.linspace(-0.4, 0.4, 3), np.linspace(0.5, 7, 4), np.linspace(-1, 1, 3),
                np.linspace(0.5, 7, 3), np.linspace(-2, 2, 3), np.linspace(0.0, 0.6, 4), np.linspace(-0.5, 0.5, 3)]
    inputs = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, len(axes))
    outputs = rng.normal(size=(inputs.shape[0], T, 2))
    bad = rng.random(inputs.shape[0]) < infeasible
    outputs[bad] = tables.INFEASIBLE
    return inputs, outputs, bad


def test_infeasible_filter_and_load(tmp_path):
    rng = np.random.default_rng(0)
    inputs, outputs, bad = _grid_table(rng, tables.FRENET, infeasible=0.2)
    # a row with a single -999 entry survives: the reference keeps rows with ANY entry != -999
    keep_partial = np.flatnonzero(~bad)[3]
    outputs[keep_partial, 2, 1] = tables.INFEASIBLE
    path = str(tmp_path / "t.npz")
    np.savez(path, inputs=inputs, outputs=outputs)
    got_in, got_out = tables.load_table(path, tables.FRENET)
    valid_ind = np.unique(np.where(outputs != -999)[0])                     # train_nmpc_frenet.py:51
    np.testing.assert_array_equal(got_in, inputs[valid_ind])
    np.testing.assert_array_equal(got_out, outputs[valid_ind])
    assert keep_partial in valid_ind and got_in.shape[0] == (~bad).sum()
    # the Cartesian trainer does not filter
    c_in, c_out = tables.load_table(path, tables.CARTESIAN)
    assert c_in.shape[0] == inputs.shape[0]
    np.savez(path, inputs=inputs, outputs=outputs[:, :, 0])
    with pytest.raises(ValueError):
        tables.load_table(path)


@pytest.mark.parametrize("kind", [tables.CARTESIAN, tables.FRENET])
def test_mirror_flatten_match_reference_statements(kind):
    rng = np.random.default_rng(1)
    inputs, outputs, _ = _grid_table(rng, kind)
    m_in, m_out = tables.mirror(inputs, outputs, kind)
    accel, deltv = outputs[:, :, 0], outputs[:, :, 1]
    cols = [inputs[:, i].flatten() for i in range(inputs.shape[1])]
    flip = {tables.CARTESIAN: (2, 3), tables.FRENET: (0, 6)}[kind]
    ref_cols = [np.concatenate((c, -c if i in flip else c), axis=0) for i, c in enumerate(cols)]   # :61-70 / :89-99
    ref_in = np.vstack(ref_cols).T                                                                   # :170 / :201
    ref_out = np.hstack([np.concatenate((accel, accel), axis=0), np.concatenate((deltv, -deltv), axis=0)])
    np.testing.assert_array_equal(m_in, ref_in)
    np.testing.assert_array_equal(tables.flatten_outputs(m_out), ref_out)
    np.testing.assert_array_equal(tables.flatten_outputs(m_out, only_onestep=True), ref_out[:, [0, 5]])


def test_bounds_and_model_card():
    rng = np.random.default_rng(2)
    inputs, outputs, _ = _grid_table(rng, tables.CARTESIAN)
    inputs, outputs = tables.mirror(inputs, outputs, tables.CARTESIAN)
    splits = [2, 1, 3, 2, 1, 1, 1]
    lower, upper, dr, R = tables.generate_bounds(inputs, splits)
    for d, n in enumerate(splits):
        b = np.sort(np.unique(inputs[:, d]))
        ind = np.linspace(start=0, stop=len(b) - 1, num=n + 1, endpoint=True, dtype=int)      # train_nmpc.py:106-133
        assert lower[d] == list(b[ind[:-1]]) and upper[d] == list(b[ind[1:]])
        assert lower[d][0] == inputs[:, d].min() and upper[d][-1] == inputs[:, d].max()
    assert R == 12 and len(dr) == 12 and dr[0] == [0] * 7 and dr[-1] == [1, 0, 2, 1, 0, 0, 0]
    # 'ij' meshgrid order: the last split dimension varies fastest
    assert dr[1] == [0, 0, 0, 1, 0, 0, 0]
    card = tables.model_card(inputs, tables.flatten_outputs(outputs), splits, 16, "gaussian")
    net = WCRBFNet.from_config(card)
    assert net.num_regions == 12 and net.in_features == 7 and net.out_features == 10
    import yaml
    yaml.safe_dump(card)          # plain Python scalars only
    with pytest.raises(ValueError):
        tables.generate_bounds(inputs, [1, 2])
