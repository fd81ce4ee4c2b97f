"""CPU: the shard rules of the device Philox state: a checkpoint restores the stream POSITION only, the shard
base stays the loading rank's; a base that is not a multiple of 4 rows is refused (draws are addressed by quads)."""
import pytest
import torch

from sdeflow_light_amd._lib import MsgmError, PhiloxState


def test_load_state_dict_keeps_the_loading_ranks_shard_base():
    r0 = PhiloxState(123, "cpu", row_base=0, n=12288)
    r1 = PhiloxState(123, "cpu", row_base=32, n=12288)
    r0.state[1] = 57                                         # rank 0 trained for a while and wrote the checkpoint
    saved = r0.state_dict()
    assert saved == {"seed": 123, "offset": 57, "row_base": 0, "elem_base": 0}
    r1.load_state_dict(saved)
    assert [int(v) for v in r1.state.tolist()] == [123, 57, 32, 32 * 12288]
    r0b = PhiloxState(999, "cpu")
    r0b.load_state_dict(saved)
    assert [int(v) for v in r0b.state.tolist()] == [123, 57, 0, 0]


@pytest.mark.parametrize("row_base,n", [(2731, 4096), (2, 2), (1, 4)])
def test_shard_base_must_be_a_multiple_of_four_rows(row_base, n):
    with pytest.raises(MsgmError):
        PhiloxState(1, "cpu", row_base=row_base, n=n)
    PhiloxState(1, "cpu", row_base=4 * row_base, n=n)       # fine
