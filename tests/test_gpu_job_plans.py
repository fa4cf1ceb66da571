"""All 113 JOB plans (tree shapes of the reference's plans.json) on the GPU vs the oracle,
over synthetic IMDB-shaped inputs (the IMDB CSVs and DuckDB are not available offline, so the
harness' DuckDB check is replaced by GPU == oracle on identical Plans)."""
import pytest

import _oracle
from pyrj import capi, job
from pyrj import plan as pl

pytestmark = pytest.mark.gpu

FX = job.load_fixture()
TABLES = job.make_tables(FX["schema"], seed=2, scale=4.0)


@pytest.fixture(scope="module")
def ctx():
    c = capi.build_context()
    yield c
    capi.destroy_context(c)


@pytest.mark.parametrize("name", sorted(FX["queries"]))
def test_job_plan(ctx, name):
    p = job.build_plan(FX["queries"][name], FX["schema"], TABLES)
    got = capi.execute(p, ctx)
    want = _oracle.execute(p)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    assert pl.sorted_rows(got) == pl.sorted_rows(want)
