"""Import-path shim: ``from audiocodec.mdctransformer import MDCTransformer`` and
``from audiocodec import psychoacoustic`` (the reference's import paths,
``audiocodec/tests/test_psychoacoustic.py:6-7``) resolve to the MI355X implementation."""
