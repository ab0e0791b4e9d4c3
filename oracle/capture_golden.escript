#!/usr/bin/env escript
%% -*- erlang -*-
%% TEST INFRASTRUCTURE, to be run by anyone who HAS Erlang/OTP and a checkout of cloudozer/BWT (it cannot run
%% in the authoring container: no erl there).  It drives the REAL reference modules over the committed inputs and
%% writes the golden vectors that tests/test_erlang_golden.py checks the oracle (and through it the GPU path) against.
%%
%%   cd <cloudozer/BWT checkout> && make            (or: erlc -I include -o ebin src/*.erl)
%%   escript <this repo>/oracle/capture_golden.escript <this repo>/tests/golden/erlang
%%
%% Output: erlang_vectors.txt, one line per input line of capture_inputs.txt:
%%   sga    -> "sga [{S,D},...]"            seeds -> "seeds no_seeds|too_many_seeds|[..]|crash"
%%   sw     -> "sw no_match|{Score,\"Cigar\"}"
%%   align  -> "align Name nothing|crash|[{{Score,\"Cigar\"},Pos}]"   (the Cigars list of worker_bwt.erl:84-97)
main([Dir]) ->
    code:add_patha("ebin"),
    {ok, Bin} = file:read_file(filename:join(Dir, "capture_inputs.txt")),
    {Meta, FM} = fm_index:get_index("GL000193.1", 1),
    {Pc,Pg,Pt,Last} = proplists:get_value(pointers, Meta),
    Shift = proplists:get_value(shift, Meta),
    {ok, Ref} = file:read_file("fm_indices/GL000193.1_p1.ref"),
    Ext = list_to_binary(lists:duplicate(200, $N)),
    Ref1 = <<Ext/binary, Ref/binary, Ext/binary>>,
    Lines = string:tokens(binary_to_list(Bin), "\n"),
    Out = [one(string:tokens(L, " "), FM, Pc, Pg, Pt, Last, Shift, Ref1) || L <- Lines],
    ok = file:write_file(filename:join(Dir, "erlang_vectors.txt"), [[O, $\n] || O <- Out]),
    io:format("~b vectors written~n", [length(Out)]).

one(["sga", Q], FM, Pc, Pg, Pt, Last, _, _) ->
    io_lib:format("sga ~w", [catch_crash(fun() -> sga:sga(FM, Pc, Pg, Pt, Last, Q) end)]);
one(["seeds", Q], FM, Pc, Pg, Pt, Last, _, _) ->
    io_lib:format("seeds ~w", [catch_crash(fun() -> bwa:find_seeds(FM, Pc, Pg, Pt, Last, Q) end)]);
one(["sw", Q, R], _, _, _, _, _, _, _) ->
    io_lib:format("sw ~p", [sw:sw(Q, R)]);
one(["align", Name, Q], FM, Pc, Pg, Pt, Last, Shift, Ref1) ->
    R = catch_crash(fun() ->
        case sga:sga(FM, Pc, Pg, Pt, Last, Q) of
            [] -> nothing;
            Seeds ->
                lists:foldl(fun({S, D}, Acc) ->                      %% worker_bwt.erl:84-97
                    RefLen = length(Q) + D,
                    Start = S - RefLen + 200,
                    <<_:Start/bytes, W:RefLen/bytes, _/binary>> = Ref1,
                    case sw:sw(Q, binary_to_list(W)) of
                        no_match -> Acc;
                        Cigar -> [{Cigar, S - RefLen + Shift} | Acc]
                    end
                end, [], Seeds)
        end end),
    io_lib:format("align ~s ~p", [Name, R]).

catch_crash(F) -> try F() catch _:_ -> crash end.
