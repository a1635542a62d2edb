"""Mirror of ``src/features/research_features.py`` (the v115 research features) backed by the HIP kernel."""
from ._frame import run_extractor


def extract_research_features(lightcurves, object_ids, metadata_df=None, verbose=True):
    """research_features.py:562-600: 40 columns per object that has rows, ``object_id`` last.  The redshift ``Z`` of
    ``metadata_df`` feeds the five luminosity columns; ids missing from it (or ``metadata_df=None``, or Z <= 0 / NaN)
    get NaN there, as the reference's ``metadata['Z'] > 0`` guard does (:552-559)."""
    return run_extractor("research", lightcurves, object_ids, metadata=metadata_df, id_last=True)
