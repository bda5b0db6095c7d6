"""Import-path alias of the reference package: ``mask_bev.*`` resolves to the MI355X-native implementation in
``mask_bev_amd`` so that callers written against the reference tree — ``from mask_bev.mask_bev_module import
MaskBevModule`` at /root/reference: train_mask_bev.py:12, the staged calls of mask_bev_figures/test_figures.py:74-76 —
run unedited.  Only the hot path of SURVEY.md §8 exists behind these names; datasets, augmentations, evaluation and
visualisation are out of scope."""
