"""sections_of(): a CascadeFilter's sections are its members -- although, like in the reference (lazy_filters.py:47-95,
970-1021), the container itself also answers numlist / denlist with the coefficients of the PRODUCT polynomial."""
import audiolazy_amd as alz
from audiolazy_amd.bank import sections_of


def test_cascade_sections_are_the_members_not_the_product():
  s_, Hz = alz.sHz(48000)
  k = alz.gammatone_erb_constants(4)[0]
  band = alz.gammatone.slaney(1000 * Hz, k * alz.erb(1000 * Hz, Hz))
  assert len(band.numlist) == 5 and len(band.denlist) == 9          # the product: order 4 over order 8
  secs = sections_of(band)
  assert len(secs) == 4
  for f, (b, a) in zip(band, secs):
    assert f.numlist == list(b) and f.denlist == list(a)
  assert len(sections_of(alz.z ** -1 + 1)) == 1
  assert sections_of(([1., 2.], [1.])) == [([1., 2.], [1.])]
  assert len(sections_of([alz.z ** -1, alz.CascadeFilter(1 - alz.z ** -1, alz.z ** -2)])) == 3
