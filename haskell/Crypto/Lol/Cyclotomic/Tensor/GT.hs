{-# LANGUAGE ConstraintKinds       #-}
{-# LANGUAGE DataKinds             #-}
{-# LANGUAGE FlexibleContexts      #-}
{-# LANGUAGE FlexibleInstances     #-}
{-# LANGUAGE GADTs                 #-}
{-# LANGUAGE KindSignatures        #-}
{-# LANGUAGE MultiParamTypeClasses #-}
{-# LANGUAGE PolyKinds             #-}
{-# LANGUAGE RankNTypes            #-}
{-# LANGUAGE ScopedTypeVariables   #-}
{-# LANGUAGE TypeFamilies          #-}
{-# LANGUAGE UndecidableInstances  #-}

-- | @GT@: a Lol 'Tensor' whose hot methods run on the MI355X library (@include/alchemy_hip.h@).
--
-- UNCOMPILED SOURCE: no Haskell toolchain (and no Lol) exists in the pipeline that produced this file, so it is
-- written against the Lol 0.7 @Tensor@ class from its published interface and checked mechanically only
-- (@tests/test_haskell_shim.py@: every method of the class is defined here, none is an @error@ stub, and every
-- foreign symbol used exists in @GT/Backend.hs@ with the header's signature).  The same call sequence, compiled
-- and tested, is @alchemy_amd/host/symmshe.hpp@ (C++).
--
-- Design.  @GT m r@ is a newtype over lol-cpp's @CT m r@ (same storable vector).  Whenever the element type is a (nested pair
-- of) @ZqBasic q Int64@ -- any cyclotomic index -- EVERY method whose result depends on the order of a basis crosses into the
-- library: @crt@, @crtInv@, @mulGCRT@, @divGCRT@ (@crtFuncs@), @twaceCRT@, @embedCRT@ (@crtExtFuncs@), and, so that the relative
-- bases agree with them, @twacePowDec@, @embedPow@, @embedDec@, @coeffs@, @powBasisPow@, @crtSetDec@, plus @mulGPow/Dec@,
-- @divGPow/Dec@, @l@, @lInv@.  The CRT slot order of an instance is internal to it (only @crtInv . crt = id@ and the ring
-- homomorphism are observable), but it must be ONE order: the instance is sound because none of these methods is delegated to
-- lol-cpp for such element types (@tests/test_haskell_shim.py@ enforces it; the identities @crt . embedPow = embedCRT . crt@ and
-- @crt . twacePowDec = twaceCRT . crt@ are GPU tests, @tests/test_tensor_ext.py@).  Order-free methods (@scalarPow@,
-- @tGaussianDec@, @gSqNormDec@, @fmapT@, @zipWithT@, @unzipT@, the entailments) and all other element types (@Double@,
-- @Complex Double@, @Int64@, @RRq@; for those the whole instance is lol-cpp's, consistently) stay on lol-cpp under 'coerce'.
--
-- Pointwise ring operations: Lol's @UCyc@ multiplies CRT-basis elements with @zipWithT (*)@, a higher-order method -- an arbitrary
-- function cannot be shipped to the GPU, and GHC rewrite rules on class methods do not fire at Lol's polymorphic call sites, so
-- @zipWithT@ (and with it the per-element @(*)@ / @(+)@ of @Cyc@) runs on lol-cpp.  That is correct in any slot order (pointwise)
-- and is not the fast path: the fast path are the batched entry points at the end of this module ('mulRelinGT', 'mulFullGT',
-- 'tunnelGT', 'modSwitchGT'), which take device-resident buffers; 'mulGT' / 'addGT' / 'subGT' are exported for callers that hold
-- @GT@ values and want the product on the device explicitly.
--
-- Use: @import Crypto.Lol.Cyclotomic.Tensor.GT@ instead of @...Tensor.CPP@ and write @GT@ for @CT@ in the
-- plaintext alias (reference examples/Arithmetic.hs:19,23; @haskell/examples/Arithmetic-GT.patch@).  Nothing in
-- @Crypto.Alchemy.*@ changes.
module Crypto.Lol.Cyclotomic.Tensor.GT ( GT, GTDispatch(..), mulGT, addGT, subGT, mulRelinGT, mulFullGT, tunnelGT, modSwitchGT ) where

import Control.Monad                          (when)
import Data.Coerce                            (coerce)
import Data.Int
import Data.IORef
import qualified Data.Map.Strict              as M
import qualified Data.Vector.Storable         as SV
import qualified Data.Vector.Storable.Mutable as SM
import Data.Word
import Foreign.C.String
import Foreign.C.Types
import Foreign.Marshal.Alloc
import Foreign.Marshal.Array
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe                       (unsafePerformIO)

import Crypto.Lol.Cyclotomic.Tensor
import Crypto.Lol.Cyclotomic.Tensor.CPP       (CT)
-- lol-cpp keeps CT's constructors in its internal module; the two marshalling functions below are the only users.
import Crypto.Lol.Cyclotomic.Tensor.CPP.Backend (CT'(..), CT(CT, ZV), zvToCT')
import Crypto.Lol.Cyclotomic.Tensor.GT.Backend
import Crypto.Lol.Prelude
import Crypto.Lol.Reflects
import Crypto.Lol.Types.Unsafe.ZqBasic        (ZqBasic)

-- | Same representation as lol-cpp's tensor.
newtype GT (m :: Factored) r = GT (CT m r)

-- | Element types the device serves: 'gtModuli' lists the RNS moduli outermost first (the nesting of
-- @PNoise2Zq@, reference Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:82-89,130); @Nothing@ = stay on lol-cpp.
class GTDispatch r where
  gtModuli :: proxy r -> Maybe [Word64]

instance (Reflects q Int64) => GTDispatch (ZqBasic q Int64) where
  gtModuli _ = Just [fromIntegral (proxy value (Proxy :: Proxy q) :: Int64)]
instance (GTDispatch a, GTDispatch b) => GTDispatch (a, b) where
  gtModuli _ = (++) <$> gtModuli (Proxy :: Proxy a) <*> gtModuli (Proxy :: Proxy b)
instance GTDispatch Int64            where gtModuli _ = Nothing
instance GTDispatch Double           where gtModuli _ = Nothing
instance GTDispatch (Complex Double) where gtModuli _ = Nothing

-- | The raw storable vector (Lol's tuple-interleaved layout) of a tensor, and back.
toVector :: SV.Storable r => GT m r -> SV.Vector r
toVector (GT (CT (CT' v))) = v
toVector (GT t@(ZV _))     = case zvToCT' t of CT' v -> v

fromVector :: SV.Storable r => SV.Vector r -> GT m r
fromVector = GT . CT . CT'

-- | One library context per (index, modulus list), created on first use and kept for the process lifetime.
{-# NOINLINE ringCache #-}
ringCache :: IORef (M.Map (Word32, [Word64], Bool) (Ptr AlchRing))
ringCache = unsafePerformIO (newIORef M.empty)

-- | @Nothing@ when q /= 1 (mod m): exactly when Lol's @crtFuncs@ has no CRT basis over the base ring.
-- With @noCRT@ the ring serves the Pow / Dec methods only (@alch_ring_create_nocrt@) and always exists.
ringFor :: Bool -> Word32 -> [Word64] -> IO (Maybe (Ptr AlchRing))
ringFor noCRT m qs = do
  cache <- readIORef ringCache
  case M.lookup (m, qs, noCRT) cache of
    Just r  -> return (Just r)
    Nothing -> alloca $ \out -> withArrayLen qs $ \n pq -> do
      rc <- (if noCRT then c_ringCreateNoCRT else c_ringCreate) m (fromIntegral n) pq out
      case rc of
        0    -> do r <- peek out
                   modifyIORef' ringCache (M.insert (m, qs, noCRT) r)
                   return (Just r)
        (-3) -> return Nothing                                  -- ALCH_E_NO_CRT
        _    -> c_lastError >>= peekCString >>= \e -> error ("alch_ring_create: " ++ e)

-- | The ring for the Pow / Dec methods of index @m@ over @r@: the CRT ring when there is one, else a no-CRT ring.
powRing :: forall m r proxy . (Fact m, GTDispatch r) => proxy (GT m r) -> Maybe (Ptr AlchRing)
powRing _ = unsafePerformIO $ case gtModuli (Proxy :: Proxy r) of
  Nothing -> return Nothing
  Just qs -> do let m = fromIntegral (proxy valueFact (Proxy :: Proxy m))
                mr <- ringFor False m qs
                maybe (ringFor True m qs) (return . Just) mr

check :: String -> CInt -> IO ()
check what rc = when (rc < 0) $ c_lastError >>= peekCString >>= \e -> error (what ++ ": " ++ e)

-- | lol-cpp's discipline: copy the input vector, let the callee mutate the copy, freeze it.
inPlace :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r
inPlace what f ring t = unsafePerformIO $ do
  mv <- SV.thaw (toVector t)
  SM.unsafeWith mv $ \p -> f ring (castPtr p) >>= check what
  fromVector <$> SV.unsafeFreeze mv

inPlace2 :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r -> GT m r
inPlace2 what f ring a b = unsafePerformIO $ do
  ma <- SV.thaw (toVector a)
  SM.unsafeWith ma $ \pa -> SV.unsafeWith (toVector b) $ \pb -> f ring (castPtr pa) (castPtr pb) >>= check what
  fromVector <$> SV.unsafeFreeze ma

-- | The @divG@ family: status 1 (@ALCH_NOT_DIVISIBLE@) is Lol's @Nothing@.
inPlaceMaybe :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> Maybe (GT m r)
inPlaceMaybe what f ring t = unsafePerformIO $ do
  mv <- SV.thaw (toVector t)
  rc <- SM.unsafeWith mv $ \p -> f ring (castPtr p)
  check what rc
  if rc == 1 then return Nothing else Just . fromVector <$> SV.unsafeFreeze mv

-- | Run @dev@ on the device when the element type has moduli, else @host@ (lol-cpp's method under the newtype).
onDevice :: forall m r a . (Fact m, GTDispatch r) => GT m r -> (Ptr AlchRing -> a) -> a -> a
onDevice t dev host = maybe host dev (powRing (Just t))

instance Tensor GT where
  type TElt GT r = (TElt CT r, GTDispatch r, SV.Storable r)

  -- ---- every basis-order-dependent method: crosses into the library ------------------------------------------------------------
  l       t = onDevice t (\ring -> inPlace "l"       c_l       ring t) (coerce (l       :: CT m r -> CT m r) t)
  lInv    t = onDevice t (\ring -> inPlace "lInv"    c_lInv    ring t) (coerce (lInv    :: CT m r -> CT m r) t)
  mulGPow t = onDevice t (\ring -> inPlace "mulGPow" c_mulGPow ring t) (coerce (mulGPow :: CT m r -> CT m r) t)
  mulGDec t = onDevice t (\ring -> inPlace "mulGDec" c_mulGDec ring t) (coerce (mulGDec :: CT m r -> CT m r) t)
  divGPow t = onDevice t (\ring -> inPlaceMaybe "divGPow" c_divGPow ring t) (coerce (divGPow :: CT m r -> Maybe (CT m r)) t)
  divGDec t = onDevice t (\ring -> inPlaceMaybe "divGDec" c_divGDec ring t) (coerce (divGDec :: CT m r -> Maybe (CT m r)) t)
  crtFuncs = crtFuncsGT
  crtExtFuncs = crtExtFuncsGT
  -- relative bases of an extension m | m': the same index rule as the CRT-slot maps above (include/alchemy_hip.h)
  twacePowDec t = between2 t (\sm bg -> outOfPlace "twacePowDec" c_twacePowDec sm bg (totOf (Proxy :: Proxy m)) t)
                             (coerce (twacePowDec :: CT m' r -> CT m r) t)
  embedPow    t = between2' t (\sm bg -> outOfPlace "embedPow" c_embedPow sm bg (totOf (Proxy :: Proxy m')) t)
                              (coerce (embedPow :: CT m r -> CT m' r) t)
  embedDec    t = between2' t (\sm bg -> outOfPlace "embedDec" c_embedDec sm bg (totOf (Proxy :: Proxy m')) t)
                              (coerce (embedDec :: CT m r -> CT m' r) t)
  coeffs      t = between2 t (\sm bg -> let n = totOf (Proxy :: Proxy m); d = totOf (Proxy :: Proxy m') `div` n
                                             v = toVector (outOfPlace "coeffs" c_coeffs sm bg (d * n) t :: GT m r)
                                         in [ fromVector (SV.slice (i * n) n v) | i <- [0 .. d - 1] ])
                             (map GT (coeffs (coerce t :: CT m' r)))
  powBasisPow   = powBasisPowGT
  crtSetDec     = crtSetDecGT
  -- zipWithT takes an arbitrary function: it cannot be shipped to the GPU (see the module header)
  zipWithT f a b = coerce (zipWithT f (coerce a :: CT m a') (coerce b :: CT m b'))

  -- ---- order-free methods: lol-cpp's implementation under the newtype -------------------------------------
  scalarPow     = coerce (scalarPow   :: r -> CT m r)
  tGaussianDec  = fmap GT . tGaussianDec
  gSqNormDec    = gSqNormDec . (coerce :: GT m r -> CT m r)
  fmapT f       = GT . fmapT f . (coerce :: GT m a -> CT m a)
  unzipT        = (\(a, b) -> (GT a, GT b)) . unzipT . (coerce :: GT m (a, b) -> CT m (a, b))
  entailIndexT  = tag $ Sub Dict
  entailEqT     = tag $ Sub Dict
  entailZTT     = tag $ Sub Dict
  entailNFDataT = tag $ Sub Dict
  entailRandomT = tag $ Sub Dict
  entailShowT   = tag $ Sub Dict
  entailModuleT = tag $ Sub Dict

-- | phi(m) as an Int.
totOf :: forall m proxy . Fact m => proxy m -> Int
totOf _ = proxy totientFact (Proxy :: Proxy m)

-- | Both Pow / Dec rings of an extension m | m' over @r@ (small, big), or @Nothing@ when @r@ stays on lol-cpp.
extRings :: forall m m' r . (Fact m, Fact m', GTDispatch r) => Proxy m -> Proxy m' -> Proxy r -> Maybe (Ptr AlchRing, Ptr AlchRing)
extRings _ _ _ = (,) <$> powRing (Proxy :: Proxy (GT m r)) <*> powRing (Proxy :: Proxy (GT m' r))

-- | Dispatch of a method @GT m' r -> a@ (big to small) / @GT m r -> a@ (small to big) on the two rings of the extension.
between2 :: forall m m' r a . (m `Divides` m', GTDispatch r) => GT m' r -> (Ptr AlchRing -> Ptr AlchRing -> a) -> a -> a
between2 _ dev host = maybe host (uncurry dev) (extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r))

between2' :: forall m m' r a . (m `Divides` m', GTDispatch r) => GT m r -> (Ptr AlchRing -> Ptr AlchRing -> a) -> a -> a
between2' _ dev host = maybe host (uncurry dev) (extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r))

-- | Out-of-place call between the two rings of an extension: the callee reads the input vector and fills a fresh vector of
-- @len@ ring-element words (nothing is retained; lol-cpp's discipline).
outOfPlace :: (SV.Storable r) => String -> (Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt)
           -> Ptr AlchRing -> Ptr AlchRing -> Int -> GT i r -> GT o r
outOfPlace what f small big len t = unsafePerformIO $ do
  out <- SM.new len
  SV.unsafeWith (toVector t) $ \pin -> SM.unsafeWith out $ \pout -> f small big (castPtr pin) (castPtr pout) >>= check what
  fromVector <$> SV.unsafeFreeze out

-- | @crtExtFuncs@ = (twaceCRT, embedCRT).  On the device exactly when 'crtFuncsGT' is (moduli present and every q = 1 mod m'):
-- both act on CRT slots and must follow the slot order of @crt@ / @crtInv@, so they are never lol-cpp's for such element types.
crtExtFuncsGT :: forall mon m m' r . (m `Divides` m', CRTrans mon r, TElt GT r) => mon (GT m' r -> GT m r, GT m r -> GT m' r)
crtExtFuncsGT =
  let m     = fromIntegral (proxy valueFact (Proxy :: Proxy m))  :: Word32
      m'    = fromIntegral (proxy valueFact (Proxy :: Proxy m')) :: Word32
      rings = unsafePerformIO $ case gtModuli (Proxy :: Proxy r) of
                Nothing -> return Nothing
                Just qs -> do big <- ringFor False m' qs             -- q = 1 mod m' implies q = 1 mod m
                              small <- maybe (return Nothing) (const (ringFor False m qs)) big
                              return ((,) <$> small <*> big)
      host  = (\(tw, em) -> (coerce tw, coerce em)) <$> (crtExtFuncs :: mon (CT m' r -> CT m r, CT m r -> CT m' r))
  in case rings of
       Nothing           -> host
       Just (small, big) -> (\_ -> ( outOfPlace "twaceCRT" c_twaceCRT small big (totOf (Proxy :: Proxy m))
                                   , outOfPlace "embedCRT" c_embedCRT small big (totOf (Proxy :: Proxy m')) )) <$> host

-- | @powBasisPow@: the relative powerful basis of m'/m as Pow-basis tensors -- unit vectors at the positions the library's
-- @coeffs@ reads first (table ALCH_EXT_COEFFS, entries [i][0]), so that @x = sum_i embed (coeffs x !! i) * powBasisPow !! i@.
powBasisPowGT :: forall m m' r . (m `Divides` m', TElt GT r, Ring r) => Tagged m [GT m' r]
powBasisPowGT = tag $ case gtModuli (Proxy :: Proxy r) of
  Nothing -> map GT (proxy powBasisPow (Proxy :: Proxy m) :: [CT m' r])
  Just _  -> unsafePerformIO $ do
    let m  = fromIntegral (proxy valueFact (Proxy :: Proxy m))  :: Word32
        m' = fromIntegral (proxy valueFact (Proxy :: Proxy m')) :: Word32
        n  = totOf (Proxy :: Proxy m)
        n' = totOf (Proxy :: Proxy m')
        d  = n' `div` n
    tab <- alloca $ \plen -> allocaArray (d * n) $ \pt -> do
             poke plen (fromIntegral (d * n))
             c_extTable m m' 1 pt plen >>= check "alch_ext_table"            -- ALCH_EXT_COEFFS
             peekArray (d * n) pt
    return [ fromVector (SV.generate n' (\k -> if k == fromIntegral (tab !! (i * n)) then one else zero)) | i <- [0 .. d - 1] ]

-- | @crtSetDec@: the relative mod-p CRT set of O_m' / O_m over the prime field @fp@ on the decoding basis, from the library's
-- host-side construction (@alch_crt_set_dec@) when @fp@ is a @ZqBasic p Int64@; lol-cpp's otherwise.  (The SET is canonical; its
-- order is the library's documented rule.  It is a list of decoding-basis vectors, not slot-indexed data, so either source would be
-- sound; the library's is used so that @decToCRT@ of reference examples/Common.hs:65-75 is reproducible from the C ABI alone.)
crtSetDecGT :: forall m m' fp . (m `Divides` m', PrimeField fp, Coprime (PToF (CharOf fp)) m', TElt GT fp) => Tagged m [GT m' fp]
crtSetDecGT = tag $ case gtModuli (Proxy :: Proxy fp) of
  Just [p] -> unsafePerformIO $ do
    let m  = fromIntegral (proxy valueFact (Proxy :: Proxy m))  :: Word32
        m' = fromIntegral (proxy valueFact (Proxy :: Proxy m')) :: Word32
        n' = totOf (Proxy :: Proxy m')
    cnt <- alloca $ \pc -> poke pc 0 >> c_crtSetDec m m' (fromIntegral p) nullPtr pc >>= check "alch_crt_set_dec" >> peek pc
    let c = fromIntegral cnt :: Int
    mv <- SM.new (c * n')
    SM.unsafeWith mv $ \pv -> alloca $ \pc -> poke pc cnt >> c_crtSetDec m m' (fromIntegral p) (castPtr pv) pc >>= check "alch_crt_set_dec"
    v <- SV.unsafeFreeze mv
    return [ fromVector (SV.slice (i * n') n' v) | i <- [0 .. c - 1] ]
  _        -> map GT (proxy crtSetDec (Proxy :: Proxy m) :: [CT m' fp])

-- | The CRTrans-monad tuple Lol asks for: (scalarCRT, mulGCRT, divGCRT, crt, crtInv).  On the device when the
-- element type has moduli AND every modulus is 1 mod m (else @alch_ring_create@ answers ALCH_E_NO_CRT, Lol's
-- @Nothing@, and lol-cpp's own 'crtFuncs' decides: it fails in the same cases).
crtFuncsGT :: forall mon m r . (CRTrans mon r, Fact m, TElt GT r)
           => mon (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)
crtFuncsGT =
  let m     = fromIntegral (proxy valueFact (Proxy :: Proxy m)) :: Word32
      mring = unsafePerformIO $ maybe (return Nothing) (ringFor False m) (gtModuli (Proxy :: Proxy r))
      host  = (\(s, mg, dg, c, ci) -> (coerce s, coerce mg, coerce dg, coerce c, coerce ci))
                <$> (crtFuncs :: mon (r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r))
  in case mring of
       Nothing   -> host
       Just ring -> (\(s, _, _, _, _) ->
                       ( s
                       , inPlace "mulGCRT" c_mulGCRT ring
                       , inPlace "divGCRT" c_divGCRT ring         -- never fails on the CRT basis
                       , inPlace "crt"     c_crt     ring
                       , inPlace "crtInv"  c_crtInv  ring )) <$> host

-- | Pointwise product / sum / difference of two tensors of the same basis on the device, for callers that hold @GT@ values
-- (Lol's own @zipWithT (*)@ stays on lol-cpp: module header).
mulGT, addGT, subGT :: forall m r . (Fact m, TElt GT r, Ring r) => GT m r -> GT m r -> GT m r
mulGT a b = onDevice a (\ring -> inPlace2 "mul" c_mul ring a b) (coerce (zipWithT (*) (coerce a :: CT m r) (coerce b :: CT m r)))
addGT a b = onDevice a (\ring -> inPlace2 "add" c_add ring a b) (coerce (zipWithT (+) (coerce a :: CT m r) (coerce b :: CT m r)))
subGT a b = onDevice a (\ring -> inPlace2 "sub" c_sub ring a b) (coerce (zipWithT (-) (coerce a :: CT m r) (coerce b :: CT m r)))


-- | @keySwitchQuadCirc hint (x * y)@ on device-resident batches: one 'c_ctMulRelin' call.
-- Arguments: ring, hint, operand buffers (2*batch CRT-basis elements each), output buffer, batch,
-- the per-limb scalar folding both toLSD and the key switch's toMSD (see include/alchemy_hip.h).
mulRelinGT :: Ptr AlchRing -> Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
mulRelinGT ring hint a b out batch spre =
  withArray spre $ \ps -> c_ctMulRelin ring hint a b out (fromIntegral batch) ps 0 >>= check "alch_ct_mul_relin"

-- | PT2CT's whole @mul_@ (@modSwitch . keySwitchQuadCirc hint . modSwitch $ x * y@, reference PT2CT.hs:172-177):
-- one 'c_ctMulFull' call; the three rings are read off the handles.
mulFullGT :: Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
mulFullGT hint a b out batch spre =
  withArray spre $ \ps -> c_ctMulFull hint a b out (fromIntegral batch) ps 0 >>= check "alch_ct_mul_full"

-- | @tunnel hint@ on device-resident batches of linear ciphertexts (SymmSHE tunnel as E runs it, reference Eval.hs:134): one
-- 'c_ctTunnel' call.  PT2CT emits @modSwitch_ .: tunnel_ hint .: modSwitch_@ (PT2CT.hs:224-229): when the input buffer's ring holds
-- only the last limbs of the tunnel's R' ring, the leading @modSwitch@ is part of this call (the added limbs are zero and skipped).
-- Arguments: tunnel handle, input buffer (2*batch CRT-basis elements over R'), output buffer (over S'), batch, toMSD's per-limb scalar.
tunnelGT :: Ptr AlchTunnel -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
tunnelGT t a out batch spre =
  withArray spre $ \ps -> c_ctTunnel t a out (fromIntegral batch) ps 0 >>= check "alch_ct_tunnel"

-- | SymmSHE @modSwitch@ on device-resident batches of linear ciphertexts (reference Eval.hs:130): up or down by whole limbs, the
-- direction read off the two buffers' rings; the trailing @modSwitch_@ of @mul_@ and @tunnel_@ when they are not fused.
modSwitchGT :: Ptr AlchBuf -> Ptr AlchBuf -> Int -> IO ()
modSwitchGT a out batch = c_ctModSwitch a out (fromIntegral batch) 0 >>= check "alch_ct_mod_switch"
