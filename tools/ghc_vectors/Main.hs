{-# LANGUAGE NoImplicitPrelude #-}
{-# LANGUAGE OverloadedStrings #-}
-- tools/ghc_vectors/Main.hs -- dumps what NO test of sdiehl/arithmetic-circuits pins and this repository therefore
-- could only derive: the coefficients createPolynomialsFFT / FFT.interpolate produce, the quotients of
-- verificationWitness[Zk], pairing's getRootOfUnity table and the aeson encodings -- for the three known-answer
-- circuits whose Bool results the reference's own tests DO hold (test/Test/QAP.hs:48-90, Example.hs:10-38,
-- bench/Circuit.hs:17-36).  Written against arithmetic-circuits v0.2.0; NOT compiled where it was written (the
-- build image has no GHC).  Run it inside a checkout of the reference (README.md beside this file):
--
--     stack runghc --package aeson -- path/to/Main.hs > ghc_vectors.json
--     python tools/ghc_vectors/check.py ghc_vectors.json          # in this repository: compares with tests/golden/
--
-- Everything goes out through the library's own ToJSON instances (src/QAP.hs:71,79,82-90;
-- src/Circuit/Arithmetic.hs:36,59,150; src/Circuit/Affine.hs:31), so the dump pins the JSON shapes as a by-product.
module Main (main) where

import Protolude

import Circuit.Affine (AffineCircuit (..))
import Circuit.Arithmetic (ArithCircuit (..), Gate (..), Wire (..), generateRoots)
import Circuit.Expr (execCircuitBuilder, deref, input, mul, add)
import Circuit.Lang (ret)
import qualified Data.Aeson as A
import Data.Aeson ((.=))
import qualified Data.ByteString.Lazy.Char8 as BL
import qualified Data.Map as Map
import Data.Pairing.BN254 (Fr, getRootOfUnity)
import Fresh (evalFresh, fresh)
import QAP

-- test/Test/QAP.hs:48-62
testArithCircuit :: ArithCircuit Fr
testArithCircuit = ArithCircuit
  [ Mul (Var (InputWire 0)) (Var (InputWire 1)) (IntermediateWire 0)
  , Mul (Var (InputWire 2)) (Var (InputWire 3)) (IntermediateWire 1)
  , Mul (Add (ConstGate 10) (Var (IntermediateWire 0))) (Var (IntermediateWire 1)) (OutputWire 0)
  ]

-- Example.hs:10-20
exampleProgram :: ArithCircuit Fr
exampleProgram = execCircuitBuilder $ do
  i0 <- fmap deref input
  i1 <- fmap deref input
  i2 <- fmap deref input
  ret (mul (mul i0 i1) (add i0 i2))

-- bench/Circuit.hs:17-24
benchProgram :: ArithCircuit Fr
benchProgram = ArithCircuit
  [ Mul (Var (InputWire 0)) (Var (InputWire 1)) (IntermediateWire 0)
  , Mul (Var (IntermediateWire 0)) (Add (Var (InputWire 0)) (Var (InputWire 2))) (OutputWire 0)
  ]

kat :: Text -> ArithCircuit Fr -> [[Fr]] -> [QapSet Fr] -> A.Value
kat name program roots assignments = A.object
  [ "name" .= name
  , "circuit" .= program                                   -- aeson shape of ArithCircuit / Gate / AffineCircuit / Wire
  , "roots" .= roots
  , "qap" .= qap                                           -- createPolynomialsFFT: every per-wire polynomial and the target
  , "assignments" .=
      [ A.object [ "assignment" .= a                       -- aeson shape of QapSet
                 , "valid" .= verifyAssignment qap a
                 , "h" .= verificationWitness qap a
                 , "delta" .= [3, 5, 7 :: Fr]
                 , "h_zk" .= verificationWitnessZk 3 5 7 qap a ]
      | a <- assignments ]
  ]
  where qap = arithCircuitToQAPFFT getRootOfUnity roots program

main :: IO ()
main = BL.putStrLn . A.encode $ A.object
  [ "library" .= ("arithmetic-circuits-0.2.0" :: Text)
  , "roots_of_unity" .= [getRootOfUnity k :: Fr | k <- [0 .. 28]]          -- pairing-1.0.0's table, k = 0 .. two-adicity
  , "cases" .=
      [ kat "test_qap_kat_fft" testArithCircuit [[1], [2], [3]]
          [ generateAssignment testArithCircuit (Map.fromList [(0, 2), (1, 3), (2, 4), (3, 5)])
          , QapSet 1 (Map.fromList [(0, 2), (1, 3), (2, 4), (3, 5)]) (Map.fromList [(0, 7), (1, 20)]) (Map.fromList [(0, 320)]) ]
      , kat "example_hs" exampleProgram (evalFresh (generateRoots (fmap (fromIntegral . (+ 1)) fresh) exampleProgram))
          [ generateAssignment exampleProgram (Map.fromList [(0, 7), (1, 5), (2, 4)]) ]
      , kat "bench_circuit" benchProgram (evalFresh (generateRoots (fromIntegral <$> fresh) benchProgram))
          [ generateAssignment benchProgram (Map.fromList [(0, 7), (1, 5), (2, 4)]) ]
      ]
  ]
